"""Diagnostic: ms per LM iteration (bind excluded) for the per-level launch solver (0) and the persistent
task-graph solver (2), B frames per launch.   python tools/time_solver.py [workload] [B ...]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-super_amd")); sys.path.insert(0, ROOT)
import torch
from super_amd import synth
from super_amd.engine import DeviceFrame, Engine
dev = torch.device("cuda", 0)
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
Bs = [int(x) for x in sys.argv[2:] if not x.startswith("-")] or [1, 8]
PATHS = (3,) if "--launches" in sys.argv else ((2,) if "--dag" in sys.argv else ((4,) if "--hybrid" in sys.argv else (3, 2, 4)))
for B in Bs:
    frames = [DeviceFrame.from_scene(synth.make_scene(seed=s, **synth.WORKLOADS[wl]), dev) for s in range(B)]
    for sp in PATHS:
        eng = Engine(dev, max_frames=B, solver_path=sp)
        for i, fr in enumerate(frames):
            eng.bind(i, fr)
        eng.run(B)
        torch.cuda.synchronize()
        ts = []
        for rep in range(5):
            for i, fr in enumerate(frames):
                eng.bind(i, fr)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            eng.run(B)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        recs = eng.records(0)
        eng.profile(True)
        eng.run(B)
        ph = eng.profile_read()
        eng.profile(False)
        solve = ph["solve"]["ms"] / max(ph["solve"]["count"], 1)
        print(f"{wl} B={B} solver_path={sp}: {min(ts) / 10:.3f} ms per iteration ({10 * B / min(ts) * 1e3:.0f} it/s), "
              f"solve phase {solve:.3f} ms, final loss {recs[-1]['loss']:.6e} status {[r['status'] for r in recs][-1]}", flush=True)
        eng.close()
