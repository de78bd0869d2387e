"""Pin the PyTorch-CPU restatement of the reference's autograd path (GraphFit) against
golden vectors recorded from the reference itself.  CPU only."""
import numpy as np
import pytest
import torch

from helpers import load_golden
from oracle import graphfit_oracle as gfo

CASES = [("s60x80_j48", "sgd"), ("s60x80_j48", "adam"), ("s60x80_j48", "sgdface"),
         ("s60x80_j48_reject", "sgd"), ("s60x80_j48_reject", "adam")]


def _opt(tag):
    return gfo.default_opt(optimizer="Adam" if tag == "adam" else "SGD", mesh_face=(tag == "sgdface"))


@pytest.mark.parametrize("name,tag", CASES)
def test_iteration0_losses_and_gradient(name, tag):
    g, sc, _ = load_golden(name)
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
    g, sc, _ = load_golden(name)
    dv = gfo.graphfit(gfo.Problem(sc), _opt(tag))
    np.testing.assert_allclose(dv, g[f"gf_{tag}_final"], rtol=0, atol=1e-10)
    assert np.abs(dv - np.eye(1, 7)).max() > 1e-7      # the optimiser moved
