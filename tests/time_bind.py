import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-super_amd")); sys.path.insert(0, ROOT)
import torch
from super_amd import synth
from super_amd.engine import DeviceFrame, Engine
dev = torch.device("cuda", 0)
sc = synth.make_scene(seed=0, **synth.WORKLOADS["C2"])
fr = DeviceFrame.from_scene(sc, dev)
for sp in (0, 1):
    eng = Engine(dev, max_frames=1, solver_path=sp)
    eng.bind(0, fr); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        eng.bind(0, fr)
    torch.cuda.synchronize()
    print("solver_path", sp, "bind ms/frame", (time.perf_counter() - t0) * 100)
    t0 = time.perf_counter()
    for _ in range(5):
        eng.run(1)
    torch.cuda.synchronize()
    print("   run ms/frame", (time.perf_counter() - t0) * 200)
