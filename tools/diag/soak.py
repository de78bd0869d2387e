"""Diagnostic: a soak of slm_run on the hybrid solver at several batch sizes -- every iteration record of every slot must be status 0
(no SLM_ITER_SOLVER_TIMEOUT / failure).  4 200 runs of 10 iterations in 75 s on the final sources of round 6: all 0."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "python-super_amd")); sys.path.insert(0, ROOT)
import numpy as np, torch
from super_amd import synth
from super_amd.engine import DeviceFrame, Engine
dev = torch.device("cuda", 0)
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
for wl, B in (("C2", 8), ("C2", 5), ("C2", 3), ("C1", 16), ("C4", 8)):
    frames = [DeviceFrame.from_scene(synth.make_scene(seed=s, **synth.WORKLOADS[wl]), dev) for s in range(B)]
    eng = Engine(dev, max_frames=B)
    t0 = time.perf_counter(); runs = 0; worst = 0
    while time.perf_counter() - t0 < secs:
        eng.bind_batch(frames)
        eng.run(B)
        torch.cuda.synchronize()
        worst = max(worst, max(r["status"] for i in range(B) for r in eng.records(i)))
        runs += 1
    print(f"{wl} B={B}: {runs} runs in {secs:.0f} s, worst iteration status {worst}, solver form {eng.plan_info(0).get('solver')}", flush=True)
    assert worst == 0
    del eng, frames
print("soak ok")
