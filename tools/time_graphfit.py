"""Diagnostic: GraphFit (Adam, 10 iterations) on one C2 frame through the C ABI."""
import sys, os, time, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-super_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from super_amd import synth
from super_amd.deform_mesh import GraphFit
from helpers import torch_frame
from super_amd import synth as _synth
sc = synth.make_scene(seed=0, **synth.WORKLOADS["C2"])
sf, inputs, new_data = torch_frame(sc)
opt = _synth.graphfit_options(optimizer="Adam")
gf = GraphFit(opt)
bf = gf._bind(0, inputs, sf, new_data)
st = torch.cuda.current_stream().cuda_stream
for _ in range(2): gf.lib.slm_gf_run(gf.h, 1, st)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10
for _ in range(n): gf.lib.slm_gf_run(gf.h, 1, st)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print("GraphFit Adam 10 iterations: %.3f ms/frame -> %.0f Adam it/s" % (dt * 1e3, 10 / dt))
