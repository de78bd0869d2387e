"""Shared test helpers: load a committed golden fixture into oracle / scene objects."""
import os
from types import SimpleNamespace

import numpy as np

from oracle import lm_oracle as orc
from super_amd import synth

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDENS = ["s60x80_j48", "s120x160_j108", "s60x80_j48_dataonly", "s60x80_j48_reject"]
GOLDENS_K = ["s60x80_j48_k6"]   # num_neighbors != 4 (6): the K-generic pair path on the multifrontal solver (round 6)


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


def torch_frame(sc, device="cuda"):
    """Reference-shaped ``sf`` / ``inputs`` / ``new_data`` objects (f64 / i64 torch tensors
    on ``device``) from a Scene -- what ``SuPer.fusion`` hands to ``LM_Solver.LM``."""
    import torch
    t = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dt)
    ed = SimpleNamespace(points=t(sc.ed_points), norms=t(sc.ed_norms), radii=t(sc.ed_radii),
                         knn_indices=t(sc.ed_knn_idx, torch.long), knn_w=t(sc.ed_knn_w),
                         num=sc.J, param_num=7 * sc.J)
    sf = SimpleNamespace(points=t(sc.sf_points), norms=t(sc.sf_norms),
                         knn_indices=t(sc.sf_knn_idx, torch.long), knn_w=t(sc.sf_knn_w),
                         ED_nodes=ed, isStable=torch.ones(sc.N, dtype=torch.bool, device=device))
    inputs = {("color", 0): torch.zeros(1, 3, sc.H, sc.W), "K": torch.from_numpy(sc.K)[None].to(device),
              "ID": torch.tensor([1])}
    new_data = SimpleNamespace(points=t(sc.tgt_points), norms=t(sc.tgt_norms),
                               index_map=t(sc.index_map, torch.long), valid=t(sc.valid, torch.bool))
    return sf, inputs, new_data


def ref_opt(opt):
    """oracle opt namespace -> adds the attributes the host mirrors read."""
    o = SimpleNamespace(**vars(opt))
    o.use_derived_gradient = True
    o.num_neighbors = 4
    o.num_ED_neighbors = 4
    return o


# option overrides of the Semantic-SuPer GraphFit variants recorded in s60x80_j48_semantic.npz
# (same table as tests/golden/make_golden.py GF_VARIANTS["semantic"])
GF_SEMANTIC_VARIANTS = {
    "soft": dict(optimizer="SGD", sf_point_plane=False, sf_soft_seg_point_plane=True, mesh_face=True,
                 sf_bn_morph=True, sf_bn_morph_weight=0.1),
    "softadam": dict(optimizer="Adam", sf_point_plane=False, sf_soft_seg_point_plane=True, mesh_face=True,
                     sf_bn_morph=True, sf_bn_morph_weight=0.1, learning_rate=1e-4),
    "hard": dict(optimizer="SGD", sf_point_plane=False, sf_hard_seg_point_plane=True),
    "morph": dict(optimizer="SGD", sf_point_plane=True, sf_bn_morph=True, sf_bn_morph_weight=1e-6),
    "clip": dict(optimizer="SGD", depth_model="raft_stereo"),
}


# option overrides of the flow-correspondence GraphFit variants recorded in s60x80_j48_corr.npz
# (same table as tests/golden/make_golden_corr.py VARIANTS)
GF_CORR_VARIANTS = {
    "corr": dict(optimizer="SGD", sf_corr=True, sf_corr_weight=0.05, sf_corr_loss_type="point-point"),
    "corradam": dict(optimizer="Adam", sf_corr=True, sf_corr_weight=0.05, sf_corr_loss_type="point-point",
                     learning_rate=1e-4),
    "corrpp": dict(optimizer="SGD", sf_corr=True, sf_corr_weight=0.5, sf_corr_loss_type="point-plane"),
    "corronly": dict(optimizer="SGD", sf_corr=True, sf_corr_weight=0.05, sf_corr_loss_type="point-point",
                     sf_point_plane=False),
}


def load_corr_golden():
    g = np.load(os.path.join(GOLDEN_DIR, "s60x80_j48_corr.npz"))
    sc = synth.Scene(H=int(g["H"]), W=int(g["W"]), K=g["K"],
                     **{k[3:]: g[k] for k in g.files if k.startswith("in_")})
    return g, sc
