"""Shared test helpers: load a committed golden fixture into oracle / scene objects."""
import os
from types import SimpleNamespace

import numpy as np

from oracle import lm_oracle as orc
from super_amd import synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDENS = ["s60x80_j48", "s120x160_j108", "s60x80_j48_dataonly", "s60x80_j48_reject"]


def load_golden(name):
    g = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
    sc = synth.Scene(H=int(g["H"]), W=int(g["W"]), K=g["K"],
                     **{k[3:]: g[k] for k in g.files if k.startswith("in_")})
    flags, w = g["opt_flags"], g["opt_weights"]
    opt = orc.default_opt(sf_point_plane=bool(flags[0]), mesh_arap=bool(flags[1]),
                          mesh_rot=bool(flags[2]), sf_point_plane_weight=float(w[0]),
                          mesh_arap_weight=float(w[1]), mesh_rot_weight=float(w[2]))
    return g, sc, opt


def coo_dense(idx, val, shape):
    out = np.zeros(tuple(int(s) for s in shape))
    np.add.at(out, (idx[0], idx[1]), val)
    return out
