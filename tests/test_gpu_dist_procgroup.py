"""The surfel-sharded modes under a REAL process group: two fresh processes, both on cuda:0 (one GPU on this box),
torch.distributed with the gloo backend (host-staged exchanges; "nccl" = RCCL cannot run two ranks on one device),
driving LM_Solver(opt, shard_surfels=True) and GraphFit(opt, shard_surfels=True) exactly as a two-GPU job would --
no injected lambdas: the collectives are the mirrors' own defaults (super_amd.dist.default_collectives), the
all-reduce runs in place on the library's exchange buffer (slm_lm_exchange_ptr)."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, mode, name, tmp):
    for p in (ROOT, os.path.join(ROOT, "python-super_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from helpers import load_golden, ref_opt, torch_frame
        g, sc, opt = load_golden(name)
        sf, inputs, new_data = torch_frame(sc)
        if mode == "lm":
            from super_amd.LM import LM_Solver
            o = ref_opt(opt)
            if name.endswith("_reject"):                       # run on past the fixture's rejected (last) iteration
                o.num_optimize_iterations = int(o.num_optimize_iterations) + 8
            lm = LM_Solver(o, shard_surfels=True)
            assert (lm.rank, lm.world) == (rank, world)
            beta = lm.LM(sf, inputs, new_data).cpu().numpy()
            recs = lm.last_records[0]
            np.savez(os.path.join(tmp, f"rank{rank}.npz"), beta=beta, loss=np.array([r["loss"] for r in recs]),
                     accepted=np.array([r["accepted"] for r in recs]), status=np.array([r["status"] for r in recs]),
                     M_grad=np.array([r["M_grad"] for r in recs]), M_loss=np.array([r["M_loss"] for r in recs]))
        else:
            from oracle import graphfit_oracle as gfo
            from super_amd.deform_mesh import GraphFit
            o = gfo.default_opt(optimizer="SGD")
            o.deform_udpate_method = "super_edg"
            gf = GraphFit(o, shard_surfels=True)
            sf.ED_nodes.triangles = torch.from_numpy(sc.ed_triangles).cuda()
            sf.ED_nodes.triangles_areas = torch.from_numpy(sc.ed_triangle_areas).cuda().double()
            dv = gf(inputs, sf, new_data).cpu().numpy()
            np.savez(os.path.join(tmp, f"rank{rank}.npz"), dv=dv)
    finally:
        dist.destroy_process_group()


def _spawn(mode, name, tmp_path):
    import torch.multiprocessing as mp
    world, port = 2, 29700 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(world, port, mode, name, str(tmp_path)), nprocs=world, join=True)
    return [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]


@pytest.mark.parametrize("name", ["s60x80_j48", "s120x160_j108"])
def test_surfel_sharded_lm_under_a_process_group(tmp_path, name):
    from helpers import load_golden
    g, _, _ = load_golden(name)
    r0, r1 = _spawn("lm", name, tmp_path)
    assert (r0["status"] == 0).all() and (r1["status"] == 0).all()
    np.testing.assert_array_equal(r0["beta"], r1["beta"])          # delta is broadcast: bit-identical parameters
    np.testing.assert_array_equal(r0["loss"], r1["loss"])
    np.testing.assert_allclose(r0["loss"], g["lm_loss"], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(r0["beta"], g["lm_beta"], rtol=0, atol=1e-4)
    assert np.abs(r0["beta"] - g["lm_beta"]).max() < 1e-7


def test_surfel_sharded_lm_matched_counts_survive_a_reject_under_a_process_group(tmp_path):
    """the record's M_grad after a rejected step (reused Jacobian pass) must be the frame's matched
    count, not world x the count -- compared with the unsharded solve over the same (extended) iterations."""
    from helpers import load_golden, ref_opt, torch_frame
    from super_amd.LM import LM_Solver
    name = "s60x80_j48_reject"
    r0, r1 = _spawn("lm", name, tmp_path)
    _, sc, opt = load_golden(name)
    o = ref_opt(opt)
    o.num_optimize_iterations = int(o.num_optimize_iterations) + 8
    lm = LM_Solver(o)
    want_beta = lm.LM(*torch_frame(sc)).cpu().numpy()
    want = lm.last_records[0]
    acc = [r["accepted"] for r in want]
    assert False in acc[:-1]
    for r in (r0, r1):
        assert list(r["accepted"]) == acc
        assert list(r["M_grad"]) == [x["M_grad"] for x in want]
        assert list(r["M_loss"]) == [x["M_loss"] for x in want]
    np.testing.assert_array_equal(r0["beta"], r1["beta"])
    np.testing.assert_allclose(r0["beta"], want_beta, rtol=0, atol=1e-9)


def test_surfel_sharded_graphfit_under_a_process_group(tmp_path):
    from helpers import load_golden
    g, _, _ = load_golden("s60x80_j48")
    r0, r1 = _spawn("gf", "s60x80_j48", tmp_path)
    np.testing.assert_array_equal(r0["dv"], r1["dv"])
    np.testing.assert_allclose(r0["dv"], g["gf_sgd_final"], rtol=0, atol=1e-9)
