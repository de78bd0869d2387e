"""Diagnostic: bench.graphfit_c4_semantic (BASELINE configs[4]'s workload on one GPU), three times."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "python-super_amd")); sys.path.insert(0, ROOT)
import torch, bench
dev = torch.device("cuda", 0)
for _ in range(3):
    r = bench.graphfit_c4_semantic(dev)
    print("c4 semantic: %.3f ms per frame (max %.3f)" % (r["ms_per_frame"], r["ms_per_frame_max"]), flush=True)
