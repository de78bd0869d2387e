"""Diagnostic: does slm_run slow down under sustained load (no idle gaps)?  n back-to-back runs vs runs separated
by idle time."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "python-super_amd")); sys.path.insert(0, ROOT)
import torch
from super_amd import synth
from super_amd.engine import DeviceFrame, Engine
dev = torch.device("cuda", 0)
B = 8
frames = [DeviceFrame.from_scene(synth.make_scene(seed=s, **synth.WORKLOADS["C2"]), dev) for s in range(B)]
eng = Engine(dev, max_frames=B)
eng.bind_batch(frames)
eng.run(B); torch.cuda.synchronize()
for n in (1, 2, 4, 8, 16, 32):
    torch.cuda.synchronize(); time.sleep(0.2)
    t0 = time.perf_counter()
    for _ in range(n):
        eng.run(B)
    torch.cuda.synchronize()
    print(f"{n:3d} runs back to back: {(time.perf_counter() - t0) * 1e3 / n:.2f} ms per run", flush=True)
for gap in (0.0, 0.002, 0.005, 0.02):
    ts = []
    for _ in range(12):
        time.sleep(gap)
        t0 = time.perf_counter(); eng.run(B); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"idle gap {gap * 1e3:.0f} ms between runs: {sum(ts[2:]) / len(ts[2:]):.2f} ms per run")
