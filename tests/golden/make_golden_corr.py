"""Golden vectors for the surfel-correspondence term of GraphFit (``opt.sf_corr``, reference
``super/deform_mesh.py:100-109`` -> ``DataLoss.autograd_forward(..., flow=...)``, ``super/loss.py:293-345``),
recorded from the REFERENCE itself (build container only).  The optical-flow NETWORK is outside the hot path;
the loss only needs its output, so ``models.optical_flow`` is a stand-in that returns a fixed smooth flow field
(recorded in the fixture).

    python tests/golden/make_golden_corr.py      ->  tests/golden/s60x80_j48_corr.npz
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "python-super_amd"))
sys.path.insert(0, HERE)

import ref_shim  # noqa: E402
import make_golden  # noqa: E402
from super_amd import synth  # noqa: E402

torch.set_num_threads(1)

VARIANTS = (
    ("corr", dict(optimizer="SGD", sf_corr=True, sf_corr_weight=0.05, sf_corr_loss_type="point-point")),
    ("corradam", dict(optimizer="Adam", sf_corr=True, sf_corr_weight=0.05, sf_corr_loss_type="point-point",
                      learning_rate=1e-4)),
    ("corrpp", dict(optimizer="SGD", sf_corr=True, sf_corr_weight=0.5, sf_corr_loss_type="point-plane")),
    ("corronly", dict(optimizer="SGD", sf_corr=True, sf_corr_weight=0.05, sf_corr_loss_type="point-point",
                      sf_point_plane=False)),
)


def main():
    ref = ref_shim.install()
    sc = synth.make_scene(N=1500, J=48, H=60, W=80, seed=6, src_border=1, tgt_border=2, tgt_holes=0.02)
    flow = synth.smooth_flow(sc.H, sc.W, 66)
    g = dict(H=sc.H, W=sc.W, K=sc.K, in_flow=flow)
    for f in ("sf_points", "sf_norms", "sf_knn_idx", "sf_knn_w", "ed_points", "ed_norms", "ed_radii", "ed_knn_idx",
              "ed_knn_w", "tgt_points", "tgt_norms", "index_map", "valid", "ed_triangles", "ed_triangle_areas"):
        if getattr(sc, f) is not None:
            g["in_" + f] = getattr(sc, f)
    orig = ref_shim.graphfit_frame

    def frame_with_flow(sc_, stable=None):
        sf, inputs, new_data, models = orig(sc_, stable)
        sf.rgb = torch.zeros(1, 3, sc_.H, sc_.W)
        models.optical_flow = lambda a, b: torch.from_numpy(flow)
        return sf, inputs, new_data, models

    ref_shim.graphfit_frame = frame_with_flow
    try:
        okw = dict(width=sc.W, height=sc.H)
        g.update(make_golden.capture_graphfit(ref, sc, okw, VARIANTS))
    finally:
        ref_shim.graphfit_frame = orig
    for tag, _ in VARIANTS:
        print(tag, {k: float(g[k]) for k in g if k.startswith(f"gf_{tag}_term_")}, "loss0", g[f"gf_{tag}_loss0"])
    path = os.path.join(HERE, "s60x80_j48_corr.npz")
    np.savez_compressed(path, **g)
    print(path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
