"""Pin the PyTorch-CPU restatement of the reference's autograd path (GraphFit) against
golden vectors recorded from the reference itself.  CPU only."""
import numpy as np
import pytest
import torch

from helpers import GF_CORR_VARIANTS, GF_SEMANTIC_VARIANTS, load_corr_golden, load_golden
from oracle import graphfit_oracle as gfo

CASES = [("s60x80_j48", "sgd"), ("s60x80_j48", "adam"), ("s60x80_j48", "sgdface"),
         ("s60x80_j48_reject", "sgd"), ("s60x80_j48_reject", "adam")]
# num_neighbors = 6 (round 6; deform_source / get_losses are K-generic in the reference: super/deform_mesh.py:198-230)
CASES += [("s60x80_j48_k6", "sgd"), ("s60x80_j48_k6", "adam"), ("s60x80_j48_k6", "sgdface")]
CASES += [("s60x80_j48_semantic", t) for t in GF_SEMANTIC_VARIANTS]   # Semantic-SuPer terms
CASES += [("s60x80_j48_corr", t) for t in GF_CORR_VARIANTS]           # flow-correspondence term (opt.sf_corr)


def _load(name):
    return load_corr_golden() if name == "s60x80_j48_corr" else load_golden(name)[:2]


def _opt(tag):
    if tag in GF_CORR_VARIANTS:
        return gfo.default_opt(**GF_CORR_VARIANTS[tag])
    if tag in GF_SEMANTIC_VARIANTS:
        return gfo.default_opt(**GF_SEMANTIC_VARIANTS[tag])
    return gfo.default_opt(optimizer="Adam" if tag == "adam" else "SGD", mesh_face=(tag == "sgdface"))


@pytest.mark.parametrize("name,tag", CASES)
def test_iteration0_losses_and_gradient(name, tag):
    g, sc = _load(name)
    pb = gfo.Problem(sc)
    dv = torch.zeros((sc.J + 1, 7), dtype=torch.float64)
    dv[:, 0] = 1.0
    dv.requires_grad_(True)
    loss, terms = gfo.total_loss(pb, dv, _opt(tag))
    np.testing.assert_allclose(float(loss.detach()), float(g[f"gf_{tag}_loss0"]), rtol=1e-10)
    for k, v in terms.items():
        if not k.startswith("_"):
            np.testing.assert_allclose(float(v.detach()), float(g[f"gf_{tag}_term_{k}"]), rtol=1e-9, atol=1e-15)
    grad, = torch.autograd.grad(loss, dv)
    grad = grad.clone()
    grad[-1] /= sc.J
    ref = g[f"gf_{tag}_grad0"]
    np.testing.assert_allclose(grad.numpy(), ref, rtol=0, atol=1e-9 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("name,tag", CASES)
def test_final_deform_verts(name, tag):
    g, sc = _load(name)
    dv = gfo.graphfit(gfo.Problem(sc), _opt(tag))
    np.testing.assert_allclose(dv, g[f"gf_{tag}_final"], rtol=0, atol=1e-10)
    assert np.abs(dv - np.eye(1, 7)).max() > 1e-7      # the optimiser moved


def test_semantic_edge_points_and_weights_are_nontrivial():
    """The semantic fixture exercises what it claims: every class has boundary pixels, the
    soft weights are strictly inside (0,1), hard matching drops some residuals, the clip drops
    some, and some (not all) mismatching surfels pass the > 15 test of the morphing term."""
    g, sc, _ = load_golden("s60x80_j48_semantic")
    pb = gfo.Problem(sc)
    E = gfo.edge_points(pb.img_seg.numpy(), pb.C)
    assert all(len(e) > 10 for e in E)
    dv = torch.zeros((sc.J + 1, 7), dtype=torch.float64)
    dv[:, 0] = 1.0
    _, sf = gfo.deform(pb, dv)
    plain, m = gfo.point_plane(pb, sf)
    soft, _ = gfo.point_plane(pb, sf, "soft")
    hard, _ = gfo.point_plane(pb, sf, "hard")
    clip, mc = gfo.point_plane(pb, sf, None, 2e-5)
    assert 0.9 * float(plain) < float(soft) < float(plain)
    assert 0 < float(hard) < float(plain)
    assert 0 < mc < m
    assert float(gfo.bn_morph(pb, sf)) > 15


def test_corr_fixture_is_nontrivial():
    """the flow moves the sample positions by a fraction of a pixel to > 1 px, some surfels leave the valid
    window or hit unmapped taps, and the correspondence term carries a visible share of the gradient"""
    g, sc = load_corr_golden()
    assert 0.5 < np.abs(sc.flow).max() < 3 and sc.flow.shape == (1, 2, sc.H, sc.W)
    pb = gfo.Problem(sc)
    dv = torch.zeros((sc.J + 1, 7), dtype=torch.float64)
    dv[:, 0] = 1.0
    _, sf = gfo.deform(pb, dv)
    _, m_pp = gfo.point_plane(pb, sf)
    _, m_c = gfo.corr_term(pb, sf)
    assert 0 < m_c < sc.N and m_c != m_pp
    d = np.abs(g["gf_corr_grad0"] - g["gf_corronly_grad0"]).max()
    assert d > 1e-3 * np.abs(g["gf_corr_grad0"]).max()
    assert np.abs(g["gf_corronly_grad0"]).max() > 1e-6
