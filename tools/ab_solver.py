"""Diagnostic: A/B of library variants selected by per-solver environment knobs, INTERLEAVED in one process.

    python tools/ab_solver.py [workload=C2] [frames=8] [rounds=9] -- NAME:K=V,K=V NAME2:K=V ...

Boxes differ by +-2 % and separate processes on one box by +-1 % (clocks / thermal state): an effect of 1-3 % cannot be
read off two `tools/time_solver.py` runs.  Here every variant gets its own Engine (the knobs that `slm_create` reads are
set in the environment while that engine is created), all engines bind the same frames, and the timed runs alternate
A B C A B C ...; reported per variant: median / min ms per LM iteration (bind excluded) and the median of the paired
differences against the FIRST variant.  Only knobs read per solver at `slm_create` can be varied this way
(SLM_ZERO_AHEAD, SLM_ZERO_AT, SLM_ZERO_WGS, SLM_HYBRID, SLM_NO_REUSE); process-wide statics cannot."""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-super_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
from super_amd import synth
from super_amd.engine import DeviceFrame, Engine

args = sys.argv[1:]
split = args.index("--") if "--" in args else len(args)
pos, specs = args[:split], args[split + 1:]
wl = pos[0] if len(pos) > 0 else "C2"
B = int(pos[1]) if len(pos) > 1 else 8
R = int(pos[2]) if len(pos) > 2 else 9
variants = []
for sp in specs or ["base:"]:
    name, _, kv = sp.partition(":")
    variants.append((name, dict(x.split("=", 1) for x in kv.split(",") if x)))
dev = torch.device("cuda", 0)
frames = [DeviceFrame.from_scene(synth.make_scene(seed=s, **synth.WORKLOADS[wl]), dev) for s in range(B)]
KNOBS = sorted({k for _, env in variants for k in env})
engs = []
for name, env in variants:
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    e = Engine(dev, max_frames=B, solver_path=0)
    for i, fr in enumerate(frames):
        e.bind(i, fr)
    e.run(B)
    engs.append(e)
for k in KNOBS:
    os.environ.pop(k, None)
torch.cuda.synchronize()
ts = np.zeros((R, len(engs)))
for r in range(R):
    for j, e in enumerate(engs):
        for i, fr in enumerate(frames):
            e.bind(i, fr)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.run(B)
        torch.cuda.synchronize()
        ts[r, j] = (time.perf_counter() - t0) * 1e3 / 10.0
base = ts[:, 0]
for j, (name, env) in enumerate(variants):
    d = ts[:, j] - base
    recs = engs[j].records(0)
    print(f"{wl} B={B} {name:14s} median {np.median(ts[:, j]):.4f} min {ts[:, j].min():.4f} ms/iteration; paired diff vs {variants[0][0]}: "
          f"median {np.median(d) * 1e3:+.1f} us (quartiles {np.percentile(d, 25) * 1e3:+.1f} .. {np.percentile(d, 75) * 1e3:+.1f}); "
          f"final loss {recs[-1]['loss']:.6e}", flush=True)
