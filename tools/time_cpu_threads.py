import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "python-super_amd")); sys.path.insert(0, ROOT)
import torch
from oracle import graphfit_oracle as gfo
from super_amd import synth
sc = synth.make_scene(seed=0, **synth.WORKLOADS["C2"])
pb = gfo.Problem(sc)
opt = gfo.default_opt(optimizer="Adam", num_optimize_iterations=3)
print("cpu_count", os.cpu_count(), "default threads", torch.get_num_threads())
for n in (8, 16, 32, 64, 128):
    torch.set_num_threads(n)
    t0 = time.perf_counter(); gfo.graphfit(pb, opt); dt = time.perf_counter() - t0
    print(n, "threads:", round(dt / 3, 3), "s per Adam iteration")
