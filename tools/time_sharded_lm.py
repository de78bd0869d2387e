"""Diagnostic (GPU box): cost of the HOST-DRIVEN surfel-sharded LM loop (four C calls and three exchanges per iteration,
super_amd/LM.py::_run_sharded) against slm_run's on-device loop, one C2 frame, world of one rank.

    python tools/time_sharded_lm.py                      # exchanges = no-ops (the loop's own overhead)
    python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 tools/time_sharded_lm.py --nccl
                                                         # exchanges = RCCL all-reduce / broadcast in place on the library's buffers
"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-super_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from helpers import ref_opt, torch_frame
from oracle import lm_oracle as orc
from super_amd import synth
from super_amd.LM import LM_Solver

nccl = "--nccl" in sys.argv
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
if nccl:
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("nccl", rank=int(os.environ.get("RANK", "0")), world_size=int(os.environ.get("WORLD_SIZE", "1")), device_id=dev)
wl = "C2"
sc = synth.make_scene(seed=0, **synth.WORKLOADS[wl])
frame = torch_frame(sc, dev)
opt = ref_opt(orc.default_opt())
plain = LM_Solver(opt)
sharded = LM_Solver(opt, shard_surfels=True) if nccl else LM_Solver(opt, rank=0, world=1, all_reduce=lambda t: None, broadcast=lambda t: None)


def timed(lm, reps=8):
    lm.LM(*frame)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        lm.LM(*frame)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts), sorted(ts)[len(ts) // 2]


a, b = timed(plain), timed(sharded)
beta_a = plain.LM(*frame)
beta_b = sharded.LM(*frame)
print(f"{wl}, one frame, 10 LM iterations incl. bind: slm_run {a[0]:.2f} ms (median {a[1]:.2f}); surfel-sharded loop, world 1, "
      f"{'RCCL exchanges in place' if nccl else 'no-op exchanges'} {b[0]:.2f} ms (median {b[1]:.2f}): "
      f"+{(b[0] - a[0]) / 10 * 1e3:.0f} us per iteration; max |beta difference| {float((beta_a - beta_b).abs().max()):.2e}")
if nccl:
    dist.destroy_process_group()
