"""Diagnostic: repeated slm_assemble on one C2 frame (flags cleared every call)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-super_amd")); sys.path.insert(0, ROOT)
import torch
from super_amd import synth, _lib
from super_amd.engine import DeviceFrame, Engine
dev = torch.device("cuda", 0)
sc = synth.make_scene(seed=0, **synth.WORKLOADS["C2"])
fr = DeviceFrame.from_scene(sc, dev)
eng = Engine(dev, max_frames=1, solver_path=1)
eng.bind(0, fr)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    _lib.check(eng.lib.slm_assemble(eng.h, 0, None, None, eng.stream), "assemble")
torch.cuda.synchronize()
