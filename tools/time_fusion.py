"""Diagnostic: repeat the fusion timing of bench.py (next_rows.surfel_fusion) for profiling."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-super_amd")); sys.path.insert(0, ROOT)
import torch
import bench
print(bench.fusion_timing(torch.device("cuda", 0)))
