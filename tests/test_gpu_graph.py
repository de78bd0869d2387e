"""HIP ED-graph construction vs the reference's goldens and the NumPy oracle,
through the C ABI.  Needs an MI355X (-m gpu)."""
from types import SimpleNamespace

import numpy as np
import pytest

from oracle import graph_oracle as gro
from test_graph_oracle import GOLD, SEM_VARIANTS, VARIANTS

pytestmark = pytest.mark.gpu


def _run(valid, index_map, points, norms, H, W, step, seg_conf=None, hard_seg=False):
    import torch
    from super_amd.graph_encoder import DirectDeformGraph
    opt = SimpleNamespace(height=H, width=W, mesh_step_size=step, method="super")
    data = SimpleNamespace(points=torch.from_numpy(points).cuda(), norms=torch.from_numpy(norms).cuda(),
                           valid=torch.from_numpy(valid).cuda(), index_map=torch.from_numpy(index_map).cuda())
    names = ("points", "norms", "radii", "edge_index", "edges_lens", "triangles", "triangles_areas")
    if seg_conf is not None:
        opt.method, opt.hard_seg, opt.mesh_face = "semantic-super", hard_seg, True
        data.seg_conf = torch.from_numpy(seg_conf).cuda()
        data.seg = torch.argmax(data.seg_conf, 1)
        names += ("seg", "seg_conf")
    gr = DirectDeformGraph(opt)(None, data)
    out = {k: getattr(gr, k).cpu().numpy() for k in names}
    out["num"] = gr.num
    return out


def _check(out, ref):
    assert out["num"] == int(ref["num"])
    np.testing.assert_array_equal(out["edge_index"], ref["edge_index"])
    np.testing.assert_array_equal(out["triangles"], ref["triangles"])
    np.testing.assert_array_equal(out["points"], ref["points"])
    np.testing.assert_array_equal(out["norms"], ref["norms"])
    np.testing.assert_allclose(out["edges_lens"], ref["edges_lens"], rtol=1e-14)
    np.testing.assert_allclose(out["radii"], ref["radii"], rtol=1e-13)
    np.testing.assert_allclose(out["triangles_areas"], ref["triangles_areas"], rtol=1e-12)
    if "seg" in ref:
        np.testing.assert_array_equal(out["seg"], ref["seg"])
        np.testing.assert_array_equal(out["seg_conf"], ref["seg_conf"])


@pytest.mark.parametrize("tag", list(VARIANTS))
def test_graph_matches_reference_goldens(tag):
    g = np.load(GOLD)
    out = _run(g["in_valid"], g["in_index_map"], g["in_points"], g["in_norms"], int(g["in_H"]), int(g["in_W"]), VARIANTS[tag])
    _check(out, {k[len(tag) + 1:]: g[k] for k in g.files if k.startswith(tag + "_")})


@pytest.mark.parametrize("tag", list(SEM_VARIANTS))
def test_semantic_graph_matches_reference_goldens(tag):
    """Node class fields and (hard_seg + mesh_face) edges / triangles across a class boundary dropped."""
    g = np.load(GOLD)
    kw = SEM_VARIANTS[tag]
    out = _run(g["in_valid"], g["in_index_map"], g["in_points"], g["in_norms"], int(g["in_H"]), int(g["in_W"]), kw["step"],
               seg_conf=g["in_seg_conf"], hard_seg=kw["hard_seg"])
    _check(out, {k[len(tag) + 1:]: g[k] for k in g.files if k.startswith(tag + "_")})


def test_semantic_graph_full_size_matches_oracle():
    from super_amd import synth
    sc = synth.make_scene(N=1000, J=12, H=480, W=640, seed=4, tgt_holes=0.2, semantic=True, num_classes=3)
    rng = np.random.default_rng(4)
    conf = sc.tgt_seg_conf.astype(np.float64) + rng.uniform(0, 0.3, sc.tgt_seg_conf.shape)
    conf /= conf.sum(1, keepdims=True)
    for step, hard in ((20, True), (11, True), (11, False)):
        out = _run(sc.valid, sc.index_map, sc.f64("tgt_points"), sc.f64("tgt_norms"), sc.H, sc.W, step, seg_conf=conf,
                   hard_seg=hard)
        ref = gro.direct_deform_graph(sc.valid, sc.index_map, sc.f64("tgt_points"), sc.f64("tgt_norms"), step,
                                      seg_conf=conf, prune_class_edges=hard)
        _check(out, ref)
        assert out["num"] > 300 and len(np.unique(out["seg"])) == 3


def test_graph_full_size_with_isolated_anchors_matches_oracle():
    """480x640 with 35 % holes (some anchors have no edge: their radius is the mean of the others) and
    the reference's default step-size range."""
    from super_amd import synth
    sc = synth.make_scene(N=1000, J=12, H=480, W=640, seed=3, tgt_holes=0.35)
    for step in (30, 13):
        out = _run(sc.valid, sc.index_map, sc.f64("tgt_points"), sc.f64("tgt_norms"), sc.H, sc.W, step)
        ref = gro.direct_deform_graph(sc.valid, sc.index_map, sc.f64("tgt_points"), sc.f64("tgt_norms"), step)
        _check(out, ref)
        assert out["num"] > 100
