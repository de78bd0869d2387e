"""Diagnostic: per-step wall time of bind + 10 LM iterations (one frame per launch) to look for periodic
stalls of the HIP runtime's launch path."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-super_amd")); sys.path.insert(0, ROOT)
import torch
from super_amd import synth
from super_amd.engine import DeviceFrame, Engine
dev = torch.device("cuda", 0)
fr = DeviceFrame.from_scene(synth.make_scene(seed=0, **synth.WORKLOADS["C2"]), dev)
eng = Engine(dev, max_frames=1, solver_path=0)
ts = []
for k in range(45):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.bind(0, fr)
    eng.run(1)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("ms per step:", [round(t, 1) for t in ts])
