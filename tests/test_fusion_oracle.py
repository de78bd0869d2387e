"""Pin the NumPy restatement of the reference's surfel fusion step against
golden vectors recorded from the reference itself.  CPU only."""
import os
from types import SimpleNamespace

import numpy as np
import pytest

from oracle import fusion_oracle as fuo

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fu_48x64.npz")

# same table as tests/golden/make_golden_fusion.py VARIANTS
VARIANTS = {
    "default": dict(),
    "tight": dict(th_dist=0.008, th_cosine_ang=0.9),
    "nomerge_new": dict(disable_merging_new_surfels=True),
    "nomerge_exist": dict(disable_merging_exist_surfels=True, th_dist=0.008),
    "noadd": dict(disable_adding_new_surfels=True, th_dist=0.008),
    "keepall": dict(disable_removing_unstable_surfels=True, th_dist=0.02),
}
STATE = ("points", "norms", "colors", "radii", "confs", "time_stamp", "isStable", "knn_indices", "knn_w", "projdata")


def load(g):
    b = {k[3:]: g[k] for k in g.files if k.startswith("in_")}
    m = fuo.Model(b["sf_points"], b["sf_norms"], b["sf_colors"], b["sf_radii"], b["sf_confs"], b["sf_time_stamp"],
                  b["sf_isStable"], b["sf_knn_idx"], b["sf_knn_w"], b["ed_points"], b["ed_radii"])
    new = SimpleNamespace(points=b["new_points"], norms=b["new_norms"], colors=b["new_colors"], radii=b["new_radii"],
                          confs=b["new_confs"], valid=b["new_valid"])
    return b, m, new


def check(m, g, prefix):
    assert len(m.points) == len(g[prefix + "points"])
    np.testing.assert_array_equal(m.isStable, g[prefix + "isStable"])
    np.testing.assert_array_equal(m.knn_indices, g[prefix + "knn_indices"])
    np.testing.assert_array_equal(m.time_stamp, g[prefix + "time_stamp"])
    np.testing.assert_allclose(m.points, g[prefix + "points"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(m.norms, g[prefix + "norms"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(m.radii, g[prefix + "radii"], rtol=1e-13)
    np.testing.assert_allclose(m.confs, g[prefix + "confs"], rtol=1e-7)
    np.testing.assert_allclose(m.colors, g[prefix + "colors"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(m.knn_w, g[prefix + "knn_w"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(m.projdata, g[prefix + "projdata"], rtol=0, atol=1e-4)


@pytest.mark.parametrize("tag", list(VARIANTS))
def test_fusion_matches_reference(tag):
    g = np.load(GOLD)
    b, m, new = load(g)
    opt = fuo.default_opt(height=int(b["H"]), width=int(b["W"]), **VARIANTS[tag])
    fuo.fuse_input_data(m, opt, b["K"], new, int(b["time"]))
    check(m, g, f"{tag}_fuse_")
    fuo.swap_stable(m, opt, int(b["time"]))
    if f"{tag}_swap_points" in g.files:
        check(m, g, f"{tag}_swap_")
    else:
        assert len(m.points) == int(g[f"{tag}_swap_count"])


def test_fixture_exercises_layers_merges_and_additions():
    g = np.load(GOLD)
    b, m, new = load(g)
    n0 = len(m.points)
    fused = g["default_fuse_isStable"]
    assert len(fused) > n0                                   # new surfels were added
    assert (b["sf_isStable"] & ~fused[:n0]).sum() > 50       # surfels merged into others and deleted
    assert (g["default_fuse_confs"][:n0] != b["sf_confs"]).sum() > 500     # new points merged into surfels
    assert len(g["tight_fuse_points"]) > len(g["default_fuse_points"])     # a tighter test adds more surfels
