"""Diagnostic: cost of slm_bind_frame with the symbolic plan cached (same coupling graph as the previous
frame in the slot) and rebuilt (coupling graph changed), and of one 10-iteration LM run."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-super_amd")); sys.path.insert(0, ROOT)
import torch
from super_amd import synth
from super_amd.engine import DeviceFrame, Engine
dev = torch.device("cuda", 0)
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
import dataclasses
import numpy as np
sc = synth.make_scene(seed=0, **synth.WORKLOADS[wl])
full = DeviceFrame.from_scene(sc, dev)
frs = [full]
# the same node graph with 3 % of the surfels missing (a different 3 % per variant): what fusion does to
# the coupled-pair list from one frame to the next
rng = np.random.default_rng(0)
for _ in range(4):
    keep = torch.from_numpy(np.sort(rng.choice(sc.N, int(0.97 * sc.N), replace=False))).to(dev)
    frs.append(dataclasses.replace(full, sf_points=full.sf_points[keep].contiguous(), sf_norms=full.sf_norms[keep].contiguous(),
                                   sf_knn_idx=full.sf_knn_idx[keep].contiguous(), sf_knn_w=full.sf_knn_w[keep].contiguous()))
eng = Engine(dev, max_frames=1, solver_path=0)
eng.bind(0, frs[0]); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    eng.bind(0, frs[0])
torch.cuda.synchronize()
print(wl, "bind, plan cached   ms/frame", (time.perf_counter() - t0) * 100)
t0 = time.perf_counter()
for i in range(10):
    eng.bind(0, frs[1 + i % 4])
torch.cuda.synchronize()
print(wl, "bind, pair list changes every frame  ms/frame", (time.perf_counter() - t0) * 100, "pairs", eng.plan_info(0)["pairs"])
t0 = time.perf_counter()
for _ in range(5):
    eng.run(1)
torch.cuda.synchronize()
print(wl, "run ms/frame", (time.perf_counter() - t0) * 200)
