"""The rows of SURVEY 8 that the LM bench's kernel trace does not contain, under one process for a rocprofv3 pass
(profiles/collect.sh):  GraphFit at C2 with one and with eight frames per launch (rows a18-a20), the Semantic-SuPer
GraphFit step at C4 (BASELINE configs[4]), depth preprocessing (f2), surfel fusion + swap (f1), the ED-graph construction
(f3) and the K-generic LM path at num_neighbors = 6.

    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -- python3 tools/profile_rows.py

Prints one JSON line with the timings of the same calls (bench.py's own functions)."""
import json
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "python-super_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch

import bench
from super_amd import synth
from super_amd.engine import DeviceFrame, Engine

dev = torch.device("cuda", 0)
out = {}
out["graphfit_gpu"] = bench.graphfit_timing(synth.WORKLOADS["C2"], dev, 8)
out["graphfit_c4_semantic"] = bench.graphfit_c4_semantic(dev)
out["next_rows"] = bench.next_row_timings(dev)
# f3: the ED-graph construction at the SuPer image size (runs once per sequence)
from types import SimpleNamespace
from super_amd.data_loader import depth_preprocessing
from super_amd.graph_encoder import DirectDeformGraph
H, W = 480, 640
K = synth.intrinsics()
vv, uu = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
depth = torch.from_numpy((0.2 * synth._surface(uu, vv, H, W, 0.3)).astype(np.float32))[None, None].to(dev)
opt = SimpleNamespace(height=H, width=W, data="superv1", load_valid_mask=False, depth_model="monodepth2", dilate_invalid_kernel=0,
                      normal_model="naive", phase="test", mesh_step_size=11, method="super", num_neighbors=4, num_ED_neighbors=4)
inputs = {"inv_K": torch.from_numpy(np.linalg.pinv(K))[None], "K": torch.from_numpy(K)[None], ("color", 0): torch.rand(1, 3, H, W, device=dev),
          "divterm": 1.0 / (2 * 0.6 * 0.6), "filename": ["000001"], ("depth", 0): depth}
data, inputs = depth_preprocessing(opt, None, inputs)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
ed = DirectDeformGraph(opt)(inputs, data)
e1.record()
torch.cuda.synchronize()
out["ed_graph_init_ms"] = e0.elapsed_time(e1)
# the K-generic LM path (num_neighbors = 6) at C2, one frame per launch
sc = synth.make_scene(seed=0, n_neighbors=6, **synth.WORKLOADS["C2"])
eng = Engine(dev, max_frames=1)
fr = DeviceFrame.from_scene(sc, dev)
for _ in range(3):
    eng.bind(0, fr)
    eng.run(1)
torch.cuda.synchronize()
eng.profile(True)
eng.bind(0, fr)
eng.run(1)
ph = eng.profile_read()
out["lm_k6_c2_b1_phase_ms"] = {k: v["ms"] / max(v["count"], 1) for k, v in ph.items()}
print(json.dumps(out), flush=True)
