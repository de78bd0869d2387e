"""Pin the NumPy oracle against golden vectors recorded from the reference itself
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from oracle import lm_oracle as orc
from helpers import GOLDENS, GOLDENS_K, coo_dense, load_golden

TOL = 1e-9   # f64 restatement vs f64 reference: only reduction order differs


@pytest.mark.parametrize("name", GOLDENS + GOLDENS_K)
@pytest.mark.parametrize("tag", ["b0", "b1"])
def test_terms_match_reference(name, tag):
    g, sc, opt = load_golden(name)
    fr = orc.Frame.from_scene(sc)
    beta = g[f"{tag}_beta"]
    P = 7 * sc.J
    if opt.sf_point_plane:
        t = orc.data_term(fr, beta, opt.sf_point_plane_weight, grad=True)
        np.testing.assert_array_equal(t.match, g[f"{tag}_match"])          # indices bit-exact
        np.testing.assert_allclose(t.r, g[f"{tag}_data_r"], rtol=0, atol=TOL)
        assert 0 < len(t.match) <= sc.N
    if opt.mesh_arap:
        np.testing.assert_allclose(orc.arap_term(fr, beta, opt.mesh_arap_weight).r,
                                   g[f"{tag}_arap_r"], rtol=0, atol=TOL)
    if opt.mesh_rot:
        np.testing.assert_allclose(orc.rot_term(beta, opt.mesh_rot_weight).r,
                                   g[f"{tag}_rot_r"], rtol=0, atol=1e-6)     # float32 term
    JtJ, jtl, M = orc.normal_equations(fr, beta, opt)
    np.testing.assert_allclose(jtl, g[f"{tag}_jtl"], rtol=0, atol=1e-8)
    loss, _ = orc.total_loss(fr, beta, opt)
    np.testing.assert_allclose(loss, float(g[f"{tag}_loss"]), rtol=1e-8)   # Rot term is float32
    if f"{tag}_jtj_nz_idx" in g.files:
        ref = coo_dense(g[f"{tag}_jtj_nz_idx"], g[f"{tag}_jtj_nz_val"], (P, P))
        np.testing.assert_allclose(JtJ, ref, rtol=0, atol=1e-7 * max(1.0, np.abs(ref).max()))
    # sparse Jacobians entry by entry
    for term, (rows, cols, vals, nrows, r, _) in orc.jacobian_coo(fr, beta, opt).items():
        key = f"{tag}_{term}_Jidx"
        if key not in g.files:
            continue
        ref = coo_dense(g[key], g[f"{tag}_{term}_Jval"], g[f"{tag}_{term}_Jshape"])
        mine = coo_dense(np.stack([rows, cols]), vals, (nrows, P))
        np.testing.assert_allclose(mine, ref, rtol=0, atol=1e-6 if term == "rot" else TOL)


@pytest.mark.parametrize("name", ["s60x80_j48"])
def test_projection_intermediates(name):
    g, sc, opt = load_golden(name)
    fr = orc.Frame.from_scene(sc)
    t = orc.data_term(fr, g["b0_beta"], 1.0)
    np.testing.assert_allclose(t.T, g["b0_T"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(t.v, g["b0_v_"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(t.u, g["b0_u_"], rtol=0, atol=1e-10)
    np.testing.assert_array_equal(t.coords, g["b0_coords"])


@pytest.mark.parametrize("name", GOLDENS + GOLDENS_K)
def test_lm_trace_matches_reference(name):
    g, sc, opt = load_golden(name)
    fr = orc.Frame.from_scene(sc)
    trace = []
    beta = orc.lm(fr, opt, trace=trace)
    assert len(trace) == len(g["lm_loss"])
    np.testing.assert_array_equal([t["accepted"] for t in trace], g["lm_accepted"])
    np.testing.assert_allclose([t["u"] for t in trace], g["lm_u"], rtol=1e-12)
    np.testing.assert_allclose([t["loss"] for t in trace], g["lm_loss"], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(beta, g["lm_beta"], rtol=0, atol=1e-6)
    assert not g["lm_accepted"].all() or name != "s60x80_j48_reject"


@pytest.mark.parametrize("name", GOLDENS + GOLDENS_K)
def test_sparse_solve_option_reproduces_the_reference_trace(name):
    """``lm(..., solve="sparse")`` (SuperLU on the block-sparse JtJ + uI -- what the full-size checks at 4 k nodes use)
    against the reference's own trace and against the dense path, delta by delta."""
    g, sc, opt = load_golden(name)
    fr = orc.Frame.from_scene(sc)
    tr_s, tr_d = [], []
    beta_s = orc.lm(fr, opt, trace=tr_s, solve="sparse")
    beta_d = orc.lm(fr, opt, trace=tr_d)
    assert len(tr_s) == len(g["lm_loss"])
    np.testing.assert_array_equal([t["accepted"] for t in tr_s], g["lm_accepted"])
    np.testing.assert_allclose([t["loss"] for t in tr_s], g["lm_loss"], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(beta_s, g["lm_beta"], rtol=0, atol=1e-6)
    for a, b in zip(tr_s, tr_d):
        scale = max(1.0, np.abs(b["delta"]).max())
        np.testing.assert_allclose(a["delta"], b["delta"], rtol=0, atol=1e-7 * scale)
    np.testing.assert_allclose(beta_s, beta_d, rtol=0, atol=1e-7)


def test_sparse_solve_refuses_an_indefinite_matrix():
    import scipy.sparse as sp
    A = sp.csr_matrix(np.diag([4.0, -1.0, 3.0]) + 0.1 * (np.ones((3, 3)) - np.eye(3)))
    with pytest.raises(np.linalg.LinAlgError):
        orc.solve_damped_sparse(A, np.ones(3), 0.0)
    x = orc.solve_damped_sparse(A, np.ones(3), 2.0)            # damped: positive definite
    np.testing.assert_allclose((A.toarray() + 2.0 * np.eye(3)) @ x, np.ones(3), atol=1e-12)


@pytest.mark.parametrize("name", ["s60x80_j48", "s120x160_j108"])
def test_update_matches_reference(name):
    g, sc, opt = load_golden(name)
    p, n, gp, gn = orc.apply_update(sc.f64("sf_points"), sc.f64("sf_norms"), sc.sf_knn_idx,
                                    sc.f64("sf_knn_w"), sc.f64("ed_points"), sc.f64("ed_norms"),
                                    g["lm_beta"])
    np.testing.assert_allclose(p, g["upd_points"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(n, g["upd_norms"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(gp, g["upd_ed_points"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(gn, g["upd_ed_norms"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("name", ["s60x80_j48", "s120x160_j108"])
def test_knn_feeder_matches_reference(name):
    g, sc, opt = load_golden(name)
    idx, w, stable, _ = orc.surfel_knn(sc.f64("sf_points"), sc.f64("ed_points"),
                                       sc.f64("ed_radii"), 4)
    np.testing.assert_array_equal(idx, g["knn_sf_idx"])
    np.testing.assert_allclose(w, g["knn_sf_w"], rtol=0, atol=1e-12)
    np.testing.assert_array_equal(stable, g["knn_sf_stable"])
    eidx, ew, _ = orc.node_knn(sc.f64("ed_points"), sc.f64("ed_radii"), 4)
    np.testing.assert_array_equal(eidx, g["knn_ed_idx"])
    np.testing.assert_allclose(ew, g["knn_ed_w"], rtol=0, atol=1e-12)
    # the generator's own KNN (inputs of the fixture) agrees with the reference feeder
    np.testing.assert_array_equal(sc.sf_knn_idx, g["knn_sf_idx"])
