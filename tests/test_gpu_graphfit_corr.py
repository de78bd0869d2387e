"""The surfel-correspondence term of GraphFit (``opt.sf_corr``: ``super/deform_mesh.py:100-109`` ->
``DataLoss.autograd_forward(..., flow=...)``, ``super/loss.py:293-345``) on the HIP path, through the C ABI,
against the golden recorded from the reference (tests/golden/make_golden_corr.py) and against torch autograd
on the oracle.  The flow network itself is the caller's: the tests hand in a ``models.optical_flow`` that
returns the recorded field.  Needs an MI355X (-m gpu).

Tolerances: the reference samples the flow with ``F.grid_sample`` on a float32 grid and back-propagates through
it in float32; the kernel reproduces the float32 sample position and blends in the same order, its derivative
of the flow is float64 -- so the flow-path share of the gradient agrees to float32 rounding (1e-7 relative),
everything else to 1e-9 like the other terms."""
from types import SimpleNamespace

import numpy as np
import pytest

from helpers import GF_CORR_VARIANTS, load_corr_golden, torch_frame
from oracle import graphfit_oracle as gfo

pytestmark = pytest.mark.gpu


def _opt(tag, **kw):
    o = gfo.default_opt(**GF_CORR_VARIANTS[tag], **kw)
    o.deform_udpate_method = "super_edg"
    return o


def _frame(sc):
    import torch
    sf, inputs, new_data = torch_frame(sc)
    sf.rgb = torch.zeros(1, 3, sc.H, sc.W, device="cuda")
    calls = []

    def optical_flow(a, b):
        calls.append((tuple(a.shape), tuple(b.shape)))
        return [torch.zeros(1, 2, sc.H, sc.W, device="cuda"), torch.from_numpy(sc.flow).cuda()]   # list: last wins

    return sf, inputs, new_data, SimpleNamespace(optical_flow=optical_flow, calls=calls)


@pytest.mark.parametrize("tag", list(GF_CORR_VARIANTS))
def test_loss_and_gradient_at_identity_match_reference(tag):
    import torch
    from super_amd.deform_mesh import GraphFit
    g, sc = load_corr_golden()
    sf, inputs, new_data, models = _frame(sc)
    gf = GraphFit(_opt(tag))
    dv = torch.zeros((sc.J + 1, 7), dtype=torch.float64, device="cuda")
    dv[:, 0] = 1.0
    terms, matched, grad = gf.loss_and_grad(inputs, sf, new_data, dv, models)
    assert models.calls == [((1, 3, sc.H, sc.W), (1, 3, sc.H, sc.W))]
    for k in ("arap_loss", "rot_loss", "point_plane_loss", "corr_loss"):
        key = f"gf_{tag}_term_{k}"
        if key in g.files:
            np.testing.assert_allclose(terms[k], float(g[key]), rtol=1e-7, atol=1e-15)
    if f"gf_{tag}_term_point_plane_loss" not in g.files:
        assert terms["point_plane_loss"] == 0.0
    np.testing.assert_allclose(sum(terms.values()), float(g[f"gf_{tag}_loss0"]), rtol=1e-7)
    ref = g[f"gf_{tag}_grad0"]
    np.testing.assert_allclose(grad.cpu().numpy(), ref, rtol=0, atol=2e-7 * np.abs(ref).max())
    assert 0 < gf.last_corr_kept < sc.N


@pytest.mark.parametrize("tag", list(GF_CORR_VARIANTS))
def test_final_deform_verts_match_reference(tag):
    from super_amd.deform_mesh import GraphFit
    g, sc = load_corr_golden()
    sf, inputs, new_data, models = _frame(sc)
    dv = GraphFit(_opt(tag))(inputs, sf, new_data, models).cpu().numpy()
    ref = g[f"gf_{tag}_final"]
    assert np.abs(ref - np.eye(1, 7)).max() > 1e-7
    np.testing.assert_allclose(dv, ref, rtol=0, atol=1e-9)          # north_star bar: 1e-4
    assert len(models.calls) == 1                                    # flow inferred once per frame (i == 0)


@pytest.mark.parametrize("tag", ["corr", "corrpp"])
def test_gradient_at_random_point_vs_autograd_oracle(tag):
    """away from identity (global row active), unstable surfels, another scene and flow, against torch autograd"""
    import torch
    from super_amd import synth
    from super_amd.deform_mesh import GraphFit
    sc = synth.make_scene(N=3000, J=48, H=60, W=80, seed=31, src_border=1, tgt_border=3, tgt_holes=0.02)
    sc.flow = synth.smooth_flow(sc.H, sc.W, 7, amp=(2.5, 1.8))
    rng = np.random.default_rng(9)
    dv0 = np.tile(np.array([1.0, 0, 0, 0, 0, 0, 0]), (sc.J + 1, 1))
    dv0 += np.concatenate([rng.normal(0, 0.01, (sc.J + 1, 4)), rng.normal(0, 0.003, (sc.J + 1, 3))], axis=1)
    opt = _opt(tag)
    stable = rng.uniform(size=sc.N) > 0.1
    pb = gfo.Problem(sc, stable=stable)
    dvt = torch.from_numpy(dv0.copy()).requires_grad_(True)
    loss, terms = gfo.total_loss(pb, dvt, opt)
    gref, = torch.autograd.grad(loss, dvt)
    gref = gref.clone()
    gref[-1] /= sc.J
    sf, inputs, new_data, models = _frame(sc)
    sf.isStable = torch.from_numpy(stable).cuda()
    gf = GraphFit(opt)
    t, matched, grad = gf.loss_and_grad(inputs, sf, new_data, torch.from_numpy(dv0).cuda(), models)
    assert matched == terms["_matched"] and gf.last_corr_kept == terms["_corr_matched"]
    assert 0 < gf.last_corr_kept < int(stable.sum())
    np.testing.assert_allclose(t["corr_loss"], float(terms["corr_loss"].detach()), rtol=1e-7)
    np.testing.assert_allclose(sum(t.values()), float(loss.detach()), rtol=1e-7)
    np.testing.assert_allclose(grad.cpu().numpy(), gref.numpy(), rtol=0, atol=2e-7 * float(gref.abs().max()))


def test_zero_flow_point_plane_corr_equals_the_unrounded_point_plane_term():
    """with a zero flow and loss type 'point-plane' the term is the point-plane residual with the FLOAT validity
    test: on surfels away from the window edge it doubles the point-plane term (a self-consistency check)"""
    import torch
    from super_amd import synth
    from super_amd.deform_mesh import GraphFit
    sc = synth.make_scene(N=2000, J=48, H=60, W=80, seed=4, src_border=6, tgt_border=2)
    sc.flow = np.zeros((1, 2, sc.H, sc.W), np.float32)
    opt = _opt("corrpp")
    opt.sf_corr_weight = opt.sf_point_plane_weight
    sf, inputs, new_data, models = _frame(sc)
    dv = torch.zeros((sc.J + 1, 7), dtype=torch.float64, device="cuda")
    dv[:, 0] = 1.0
    t, matched, _ = GraphFit(opt).loss_and_grad(inputs, sf, new_data, dv, models)
    assert matched == sc.N
    np.testing.assert_allclose(t["corr_loss"], t["point_plane_loss"], rtol=1e-12)


def test_surfel_sharded_ranks_reproduce_the_single_gpu_solve():
    import torch
    from super_amd.deform_mesh import GraphFit
    g, sc = load_corr_golden()
    sf, inputs, new_data, models = _frame(sc)
    opt = _opt("corr")
    world = 3
    ranks = [GraphFit(opt, rank=r, world=world, all_reduce=lambda t: None) for r in range(world)]
    for gf in ranks:
        gf.bind(inputs, sf, new_data, models)
    bufs = [torch.empty((sc.J + 1) * 7 + 10, dtype=torch.float64, device="cuda") for _ in ranks]
    for _ in range(opt.num_optimize_iterations):
        for gf in ranks:
            gf.eval_morph()
            gf.eval_losses()
        tot = torch.stack([gf.get_partial(b) for gf, b in zip(ranks, bufs)]).sum(0)
        for gf in ranks:
            gf.set_partial(tot)
            gf.step()
    dvs = [gf.deform_verts().cpu().numpy() for gf in ranks]
    for d in dvs[1:]:
        np.testing.assert_array_equal(d, dvs[0])
    np.testing.assert_allclose(dvs[0], g["gf_corr_final"], rtol=0, atol=1e-9)


def test_missing_flow_fails_loudly():
    import torch
    from super_amd.deform_mesh import GraphFit
    g, sc = load_corr_golden()
    sf, inputs, new_data, _ = _frame(sc)
    with pytest.raises(ValueError, match="optical_flow"):
        GraphFit(_opt("corr"))(inputs, sf, new_data, None)
    o = _opt("corr")
    o.sf_corr_match_renderimg = True
    with pytest.raises(NotImplementedError):
        GraphFit(o)
