import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import conftest  # noqa
import numpy as np, torch
from helpers import load_golden, torch_frame, ref_opt
from oracle import lm_oracle as orc
from super_amd.LM import LM_Solver
g, sc, opt = load_golden("s60x80_j48")
sf, inputs, new_data = torch_frame(sc)
for flags in [(1,1,1),(1,0,0),(0,1,0),(0,0,1)]:
    o = orc.default_opt(sf_point_plane=bool(flags[0]), mesh_arap=bool(flags[1]), mesh_rot=bool(flags[2]))
    lm = LM_Solver(ref_opt(o))
    beta = g["b1_beta"]
    jtj, jtl = lm.prepareCostTerm(sf, inputs, new_data, torch.from_numpy(beta).cuda(), grad=True)
    fr = orc.Frame.from_scene(sc)
    J2, l2, _ = orc.normal_equations(fr, beta, o)
    d = np.abs(jtl.cpu().numpy().reshape(-1) - l2)
    bad = np.nonzero(d > 1e-8)[0]
    print(flags, "max jtl diff", d.max(), "bad idx%7:", np.bincount(bad % 7, minlength=7), "jtj diff", np.abs(jtj.cpu().numpy()-J2).max())
    if flags == (0,0,1) and len(bad):
        j = bad[0] // 7
        q = beta[j,:4].astype(np.float32)
        print("node", j, "q", q, "hip jtl", jtl.cpu().numpy().reshape(-1)[7*j:7*j+4], "oracle", l2[7*j:7*j+4])
        s = np.float32(0)
        for c in range(4): s = np.float32(s + np.float32(q[c]*q[c]))
        print("seq s", repr(s), "np sum", repr((q*q).sum(dtype=np.float32)), "pair", repr(np.float32(np.float32(q[0]*q[0]+q[1]*q[1]) + np.float32(q[2]*q[2]+q[3]*q[3]))))
