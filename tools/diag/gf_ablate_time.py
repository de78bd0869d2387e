"""Diagnostic: time slm_gf_run (C2, Adam, 1 and 8 frames per launch) against library variants (SLM_LIB), one subprocess each."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys
ROOT = %r
sys.path.insert(0, os.path.join(ROOT, "python-super_amd")); sys.path.insert(0, ROOT)
import torch, bench
from super_amd import synth
from super_amd.deform_mesh import GraphFit
device = torch.device("cuda", 0)
dims = synth.WORKLOADS[os.environ.get("GF_WL", "C2")]
opt = synth.graphfit_options(optimizer="Adam")
scs = [synth.make_scene(seed=s, **dims) for s in range(8)]
for n in (1, 8):
    gf = GraphFit(opt, max_frames=n)
    keep = [gf._bind(i, *bench._reorder(bench._graphfit_frames(scs[i], device))) for i in range(n)]
    med, mx = bench._time_gf_run(gf, n, device, reps=20)
    print(os.environ.get("SLM_LIB", "default"), n, "frames: %%.3f ms per run (max %%.3f), %%.1f us per iteration" %% (med, mx, med * 100), flush=True)
    del keep, gf
''' % ROOT
for lib in sys.argv[1:]:
    env = dict(os.environ)
    if lib != "default":
        env["SLM_LIB"] = lib
    subprocess.run([sys.executable, "-c", CHILD], env=env)
