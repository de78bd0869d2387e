"""Diagnostic: one damped solve of a small frame through the per-level launches (solver_path 0) and the
task-graph kernel (2); compares the factor tiles, the inverses, the vectors and delta buffer by buffer."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-super_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
from super_amd import _lib, synth
from super_amd.engine import DeviceFrame, Engine
dev = torch.device("cuda", 0)
kw = dict(N=1500, J=48, H=60, W=80, seed=1, src_border=4, tgt_border=6)
if len(sys.argv) > 1:
    kw.update(J=int(sys.argv[1]), N=int(sys.argv[2]), H=int(sys.argv[3]), W=int(sys.argv[4]))
sc = synth.make_scene(**kw)
bufs = {}
for sp in (0, 2):
    e = Engine(dev, solver_path=sp, data_path=2)
    e.bind(0, DeviceFrame.from_scene(sc, dev))
    print(sp, e.plan_info(0))
    d = torch.zeros(7 * sc.J, dtype=torch.float64, device=dev)
    st = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(e.lib.slm_solve(e.h, 0, 0.5, d.data_ptr(), st.data_ptr(), e.stream), "solve")
    out = {}
    for what, name in enumerate(("tiles", "vec", "linv", "delta")):
        n = C.c_int64(0)
        _lib.check(e.lib.slm_debug_read(e.h, 0, what, None, 0, C.byref(n), e.stream), "dbg")
        a = np.zeros(n.value)
        _lib.check(e.lib.slm_debug_read(e.h, 0, what, a.ctypes.data_as(C.c_void_p), n.value, C.byref(n), e.stream), "dbg")
        out[name] = a
    out["status"] = int(st.item())
    bufs[sp] = out
print("status", bufs[0]["status"], bufs[2]["status"])
for name in ("linv", "tiles", "vec", "delta"):
    a, b = bufs[0][name], bufs[2][name]
    blk = 4096 if name in ("linv", "tiles") else 64
    nb = len(a) // blk
    bad = [(i, float(np.abs(a[i * blk:(i + 1) * blk] - b[i * blk:(i + 1) * blk]).max())) for i in range(nb)]
    worst = [(i, round(v, 12)) for i, v in bad if v > 1e-9 * max(1.0, np.abs(a).max())]
    print(f"{name}: {nb} blocks of {blk}, max|a| {np.abs(a).max():.3e}, blocks that differ: {len(worst)} {worst[:40]}")
