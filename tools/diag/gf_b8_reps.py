"""Diagnostic: per-repetition time of slm_gf_run at 1 and 8 frames per launch (C2, Adam), the way bench.graphfit_timing binds them."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "python-super_amd")); sys.path.insert(0, ROOT)
import torch
import bench
from super_amd import synth
from super_amd.deform_mesh import GraphFit
device = torch.device("cuda", 0)
dims = synth.WORKLOADS["C2"]
opt = synth.graphfit_options(optimizer="Adam")
scs = [synth.make_scene(seed=s, **dims) for s in range(8)]
out = {}
for n in (1, 8, 8):
    gf = GraphFit(opt, max_frames=n)
    keep = [gf._bind(i, *bench._reorder(bench._graphfit_frames(scs[i], device))) for i in range(n)]
    st = torch.cuda.current_stream(device).cuda_stream
    for _ in range(2):
        gf.lib.slm_gf_run(gf.h, n, st)
    torch.cuda.synchronize(device)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(31)]
    evs[0].record()
    for r in range(30):
        gf.lib.slm_gf_run(gf.h, n, st)
        evs[r + 1].record()
    torch.cuda.synchronize(device)
    ms = [round(a.elapsed_time(b), 3) for a, b in zip(evs, evs[1:])]
    print(n, "frames per launch, ms per run:", ms, flush=True)
    del keep, gf
