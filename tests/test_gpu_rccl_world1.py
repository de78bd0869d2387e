"""The RCCL code path, executed.  This box has ONE GPU, so the process group has one rank -- but it
is a real ``init_process_group("nccl", device_id=...)`` group, and every collective the multi-GPU job issues runs
through RCCL on device memory:

* ``LM_Solver(opt, shard_surfels=True)``: per LM iteration the all-reduce of the pair blocks, the broadcast of delta
  and the all-reduce of the loss partials, IN PLACE on the library's hipMalloc'd exchange buffer
  (``slm_lm_exchange_ptr`` -> ``super_amd.dist.device_view`` -> ``default_collectives``);
* ``GraphFit(opt, shard_surfels=True)``: the all-reduce of the partial gradient / loss terms;
* ``all_gather_betas`` (the end-of-frame exchange of the frame-sharded mode) on device tensors;
* ``bench.py --gpus 1`` started by ``python -m torch.distributed.run --nproc-per-node 1`` exactly as the driver
  starts the N-GPU runs (launcher first, GPU touched only inside the rank).

Results must equal the reference's goldens (the single-rank sums are the whole sums)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, port, tmp):
    for p in (ROOT, os.path.join(ROOT, "python-super_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        assert dist.get_backend() == "nccl"
        from helpers import load_golden, ref_opt, torch_frame
        from super_amd import _lib
        from super_amd.LM import LM_Solver
        from super_amd.dist import all_gather_betas, default_collectives
        out = {}
        # ---- the collectives really are RCCL on device tensors
        ar, bc = default_collectives()
        t = torch.arange(8, dtype=torch.float64, device=dev)
        ar(t)
        bc(t)
        assert torch.equal(t.cpu(), torch.arange(8, dtype=torch.float64))
        # ---- surfel-sharded LM: exchanges in place on the library's buffer
        g, sc, opt = load_golden("s120x160_j108")
        sf, inputs, new_data = torch_frame(sc)
        lm = LM_Solver(ref_opt(opt), shard_surfels=True)
        assert lm.sharded and (lm.rank, lm.world) == (0, 1)
        calls = {"ar": 0, "bc": 0, "ptrs": set()}
        ar0, bc0 = lm._all_reduce, lm._broadcast

        def ar1(x):
            assert x.is_cuda
            calls["ar"] += 1
            calls["ptrs"].add(x.data_ptr())
            ar0(x)

        def bc1(x):
            assert x.is_cuda
            calls["bc"] += 1
            bc0(x)
        lm._all_reduce, lm._broadcast = ar1, bc1
        beta = lm.LM(sf, inputs, new_data)
        h = next(iter(lm._solvers.values()))
        import ctypes as C
        ptr, n = C.c_void_p(), C.c_int64(0)
        _lib.check(lm.lib.slm_lm_exchange_ptr(h, 0, _lib.SLM_X_PAIR_BLOCKS, C.byref(ptr), C.byref(n)), "ptr")
        assert ptr.value in calls["ptrs"]                       # the all-reduce ran on the library's own memory
        n_it = int(opt.num_optimize_iterations)
        assert calls["ar"] == 2 * n_it and calls["bc"] == n_it
        recs = lm.last_records[0]
        out["lm_beta"] = beta.cpu().numpy()
        out["lm_loss"] = np.array([r["loss"] for r in recs])
        out["lm_status"] = np.array([r["status"] for r in recs])
        # ---- end-of-frame all-gather of the frame-sharded mode (device tensors, RCCL)
        local = torch.stack([beta, beta + 1.0])
        full = all_gather_betas(local, 2)
        assert full.is_cuda and torch.equal(full, local)
        # ---- surfel-sharded GraphFit
        from oracle import graphfit_oracle as gfo
        from super_amd.deform_mesh import GraphFit
        g2, sc2, _ = load_golden("s60x80_j48")
        sf2, inputs2, new2 = torch_frame(sc2)
        o = gfo.default_opt(optimizer="SGD")
        o.deform_udpate_method = "super_edg"
        gf = GraphFit(o, shard_surfels=True)
        assert gf.sharded
        sf2.ED_nodes.triangles = torch.from_numpy(sc2.ed_triangles).cuda()
        sf2.ED_nodes.triangles_areas = torch.from_numpy(sc2.ed_triangle_areas).cuda().double()
        out["gf_dv"] = gf(inputs2, sf2, new2).cpu().numpy()
        np.savez(os.path.join(tmp, "rank0.npz"), **out)
    finally:
        dist.destroy_process_group()


def test_rccl_collectives_run_in_place_on_the_library_buffers(tmp_path):
    import torch.multiprocessing as mp
    from helpers import load_golden
    port = 29900 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(port, str(tmp_path)), nprocs=1, join=True)
    r = np.load(tmp_path / "rank0.npz")
    g, _, _ = load_golden("s120x160_j108")
    assert (r["lm_status"] == 0).all()
    np.testing.assert_allclose(r["lm_loss"], g["lm_loss"], rtol=1e-6, atol=1e-12)
    assert np.abs(r["lm_beta"] - g["lm_beta"]).max() < 1e-7
    g2, _, _ = load_golden("s60x80_j48")
    np.testing.assert_allclose(r["gf_dv"], g2["gf_sgd_final"], rtol=0, atol=1e-9)


def test_bench_under_torchrun_with_one_rank_uses_rccl():
    """The driver's N-GPU command line with N = 1: rendezvous, nccl group, barrier, all-gather of beta, MAX over ranks."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["OMP_NUM_THREADS"] = "4"
    port = 29300 + (os.getpid() % 500)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2",
           "--warmup", "1", "--workload", "tiny", "--frames-per-gpu", "3", "--no-cpu-baseline", "--no-latency-b1"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0
    d = dict(out["distributed"])
    per_rank = d.pop("per_rank_step_ms")                 # the extra all-gather of the per-rank step statistics ran over RCCL
    assert d == {"backend": "nccl", "world": 1, "launched_by": "torchrun"}
    assert len(per_rank["ranks"]) == 1 and per_rank["ranks"][0][0] > 0
    assert out["lm_iterations_ok_frame0"] == 10
