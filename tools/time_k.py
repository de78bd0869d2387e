"""Diagnostic: ms per LM iteration at another num_neighbors (K-generic pair path on the multifrontal solver, or the
block-banded compatibility path with SLM solver_path 1).   python tools/time_k.py [workload=C2] [K=6] [B ...]"""
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
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6
Bs = [int(x) for x in sys.argv[3:] if not x.startswith("-")] or [1, 8]
for B in Bs:
    frames = [DeviceFrame.from_scene(synth.make_scene(seed=s, n_neighbors=K, **synth.WORKLOADS[wl]), dev) for s in range(B)]
    for sp in ((1, 0) if "--band" in sys.argv else (0,)):
        eng = Engine(dev, max_frames=B, solver_path=sp)
        for i, fr in enumerate(frames):
            eng.bind(i, fr)
        eng.run(B)
        torch.cuda.synchronize()
        ts, tb = [], []
        for rep in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i, fr in enumerate(frames):
                eng.bind(i, fr)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            eng.run(B)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t1) * 1e3)
            tb.append((t1 - t0) * 1e3)
        recs = eng.records(0)
        eng.profile(True)
        eng.run(B)
        ph = eng.profile_read()
        eng.profile(False)
        per = {k: round(v["ms"] / max(v["count"], 1), 3) for k, v in ph.items()}
        print(f"{wl} K={K} B={B} solver_path={sp} form={eng.lib.slm_debug_last_solver_form(eng.h)}: {min(ts) / 10:.3f} ms per iteration "
              f"({10 * B / min(ts) * 1e3:.0f} it/s), bind {min(tb) / B:.3f} ms per frame, phases {per}, final loss {recs[-1]['loss']:.6e} "
              f"status {[r['status'] for r in recs][-1]}", flush=True)
        eng.close()
