"""Eight ranks rehearsed on the one GPU this box has.

No 8-GPU node has been available to any round, so the first real ``SCALE`` run would also be the first time eight
ranks of ``bench.py`` exist at once.  This test starts the driver's own command line,

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus 8 --steps 3 --warmup 1

with ``BENCH_SHARE_GPU=1`` (every rank on cuda:0) and ``BENCH_DIST_BACKEND=gloo`` (RCCL cannot put eight ranks on one
device), fresh child processes only, launcher started before the children touch the GPU.  Eight ranks x 8 C2 frames:
eight persistent task-graph launches share one GPU -- exactly what the wall-clock deadline of a task's wait is for.
Asserted: exit code 0, ``distributed.world == 8``, no iteration of any rank ended in ``SLM_ITER_SOLVER_TIMEOUT`` (or any
other failure), the host CPU of all ranks together stays inside the box's 16-CPU quota, and the betas rank 0 holds after
the end-of-frame all-gather equal what a single rank computes for the same global frames (first and last rank's shares
recomputed here; 1e-9: the data term's LDS merge adds in arrival order).  No scaling number is asked for or reported.
Reference: the frame-sharded mode of SURVEY 8e / BASELINE configs[3]."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORLD, B = 8, 8


def test_eight_ranks_share_one_gpu_and_gather_the_single_rank_betas(tmp_path):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.update(BENCH_SHARE_GPU="1", BENCH_DIST_BACKEND="gloo", OMP_NUM_THREADS="2",
               BENCH_DUMP_BETAS=str(tmp_path / "betas.npy"))
    port = 29500 + (os.getpid() % 400)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(WORLD), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(WORLD), "--steps", "3",
           "--warmup", "1"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines                      # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == WORLD and out["steps"] == 3 and out["value"] > 0
    d = dict(out["distributed"])
    per_rank = d.pop("per_rank_step_ms")                 # (round 5) every rank's min / median / max step and its wall time
    assert d == {"backend": "gloo", "world": WORLD, "launched_by": "torchrun"}
    assert per_rank["columns"] == ["min", "median", "max", "timed_region_wall_ms"] and len(per_rank["ranks"]) == WORLD
    assert all(0 < r[0] <= r[1] <= r[2] and r[3] > 0 for r in per_rank["ranks"]), per_rank
    assert out["config"]["frames_per_gpu"] == B and out["config"]["global_frames"] == WORLD * B
    assert out["worst_iter_status_all_ranks"] == 0, out["worst_iter_status_all_ranks"]   # 3 would be SLM_ITER_SOLVER_TIMEOUT
    assert out["lm_iterations_ok_frame0"] == 10
    assert "TIMEOUT" not in p.stderr.upper()
    quota = out["host"].get("cpu_quota") or 16.0
    busy = out["host"]["cpu_cores_busy_all_ranks"]
    print(f"8 ranks on one GPU: {out['value']:.0f} LM it/s aggregate, {out['ms_per_step']:.1f} ms per step, "
          f"host CPUs busy over all ranks {busy:.2f} (quota {quota})")
    assert busy <= min(16.0, quota), busy

    got = np.load(tmp_path / "betas.npy")
    assert got.shape[0] == WORLD * B
    import torch
    from super_amd import synth
    from super_amd.engine import DeviceFrame, Engine
    dev = torch.device("cuda", 0)
    eng = Engine(dev, max_frames=B, num_iterations=10)
    for rank in (0, WORLD - 1):
        frames = [DeviceFrame.from_scene(synth.make_scene(seed=rank * B + i, **synth.WORKLOADS["C2"]), dev) for i in range(B)]
        eng.bind_batch(frames)
        eng.run(B)
        for i in range(B):
            want = eng.beta(i).cpu().numpy()
            err = float(np.abs(got[rank * B + i] - want).max())
            assert err < 1e-9, (rank, i, err)
    eng.close()
