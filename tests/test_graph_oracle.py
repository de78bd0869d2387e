"""Pin the NumPy restatement of the reference's ED-graph construction against
golden vectors recorded from the reference itself.  CPU only."""
import os

import numpy as np
import pytest

from oracle import graph_oracle as gro

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gr_60x80.npz")
VARIANTS = {"step6": 6, "step9": 9, "step4": 4}
SEM_VARIANTS = {"sem6": dict(step=6, hard_seg=False), "hard6": dict(step=6, hard_seg=True),
                "hard4": dict(step=4, hard_seg=True)}


def check(out, g, tag):
    assert out["num"] == int(g[f"{tag}_num"])
    np.testing.assert_array_equal(out["edge_index"], g[f"{tag}_edge_index"])
    np.testing.assert_array_equal(out["triangles"], g[f"{tag}_triangles"])
    np.testing.assert_array_equal(out["points"], g[f"{tag}_points"])
    np.testing.assert_array_equal(out["norms"], g[f"{tag}_norms"])
    np.testing.assert_allclose(out["edges_lens"], g[f"{tag}_edges_lens"], rtol=1e-14)
    np.testing.assert_allclose(out["radii"], g[f"{tag}_radii"], rtol=1e-13)
    np.testing.assert_allclose(out["triangles_areas"], g[f"{tag}_triangles_areas"], rtol=1e-12)
    if f"{tag}_seg" in g.files:
        np.testing.assert_array_equal(out["seg"], g[f"{tag}_seg"])
        np.testing.assert_array_equal(out["seg_conf"], g[f"{tag}_seg_conf"])


@pytest.mark.parametrize("tag", list(SEM_VARIANTS))
def test_semantic_graph_matches_reference(tag):
    g = np.load(GOLD)
    kw = SEM_VARIANTS[tag]
    out = gro.direct_deform_graph(g["in_valid"], g["in_index_map"], g["in_points"], g["in_norms"], kw["step"],
                                  seg_conf=g["in_seg_conf"], prune_class_edges=kw["hard_seg"])
    check(out, g, tag)
    if kw["hard_seg"]:
        s = out["seg"]
        assert (s[out["edge_index"][0]] == s[out["edge_index"][1]]).all()
        assert out["edge_index"].shape[1] < g["sem6_edge_index"].shape[1] or kw["step"] != 6
        assert len(np.unique(s)) == 3


@pytest.mark.parametrize("tag", list(VARIANTS))
def test_graph_matches_reference(tag):
    g = np.load(GOLD)
    out = gro.direct_deform_graph(g["in_valid"], g["in_index_map"], g["in_points"], g["in_norms"], VARIANTS[tag])
    check(out, g, tag)
    assert out["num"] > 20 and out["edge_index"].shape[1] > out["num"]
