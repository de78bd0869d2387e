"""``--num_neighbors`` other than 4: a documented tunable of the reference (``README.md:175``,
``options.py:49``; its code is K-generic: ``super/loss.py:213-220``, ``super/utils.py:30-36``, ``super/nodes.py:170-191``).
The device path for K != 4 is the per-entry-atomics data term (``k_data_grad<K>`` / ``k_data_loss<K>``) with the
block-banded float64 solve -- what ``slm_data_path = 1`` runs for K = 4 --, ``Surfels.update`` and the KNN feeder with
runtime K.  Pinned by ``tests/golden/s60x80_j48_k6.npz``, recorded from the reference at ``num_neighbors = 6``
(``tests/golden/make_golden.py``): match set and tap rows bit-exact, residuals 1e-9, JtJ / jtl against the reference's
own sparse Jacobian, the ten-iteration LM trace, ``update`` and the feeder's tables.  Through the C ABI, on an MI355X."""
import numpy as np
import pytest

from helpers import GOLDENS_K, coo_dense, load_golden, ref_opt, torch_frame
from oracle import lm_oracle as orc

pytestmark = pytest.mark.gpu


def _opt(opt, K):
    o = ref_opt(opt)
    o.num_neighbors = K
    return o


def _solver(opt, K, **kw):
    from super_amd.LM import LM_Solver
    o = _opt(opt, K)
    for k, v in kw.items():
        setattr(o, k, v)
    return LM_Solver(o)


@pytest.mark.parametrize("name", GOLDENS_K)
@pytest.mark.parametrize("tag", ["b0", "b1"])
def test_assemble_matches_reference_golden(name, tag):
    import torch
    g, sc, opt = load_golden(name)
    K = sc.sf_knn_idx.shape[1]
    assert K == 6
    sf, inputs, new_data = torch_frame(sc)
    lm = _solver(opt, K)
    beta = torch.from_numpy(g[f"{tag}_beta"]).cuda()
    jtj, jtl = lm.prepareCostTerm(sf, inputs, new_data, beta, grad=True)
    loss = lm.prepareCostTerm(sf, inputs, new_data, beta, grad=False)
    np.testing.assert_allclose(jtl.cpu().numpy().reshape(-1), g[f"{tag}_jtl"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(float(loss), float(g[f"{tag}_loss"]), rtol=1e-8)
    P = 7 * sc.J
    ref = coo_dense(g[f"{tag}_jtj_nz_idx"], g[f"{tag}_jtj_nz_val"], (P, P))
    np.testing.assert_allclose(jtj.cpu().numpy(), ref, rtol=0, atol=1e-7 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("name", GOLDENS_K)
def test_match_set_and_residuals_bit_exact_indices(name):
    import torch
    from super_amd import _lib
    from super_amd.LM import _dev_ptr, _stream_ptr
    g, sc, opt = load_golden(name)
    K = sc.sf_knn_idx.shape[1]
    sf, inputs, new_data = torch_frame(sc)
    lm = _solver(opt, K)
    h = lm._handle()
    for tag in ("b0", "b1"):
        bf = lm._bind(h, 0, sf, inputs, new_data)
        beta = torch.from_numpy(g[f"{tag}_beta"]).cuda()
        st = _stream_ptr(bf.device)
        _lib.check(lm.lib.slm_set_beta(h, 0, _dev_ptr(beta), st), "set_beta")
        r = torch.empty(sc.N, dtype=torch.float64, device="cuda")
        m = torch.empty(sc.N, dtype=torch.uint8, device="cuda")
        taps = torch.empty((sc.N, 4), dtype=torch.int32, device="cuda")
        _lib.check(lm.lib.slm_data_residuals(h, 0, _dev_ptr(r), _dev_ptr(m), _dev_ptr(taps), st), "resid")
        match = np.nonzero(m.cpu().numpy())[0]
        np.testing.assert_array_equal(match, g[f"{tag}_match"])                 # bit-exact
        np.testing.assert_allclose(r.cpu().numpy()[match], g[f"{tag}_data_r"], rtol=0, atol=1e-9)
        t = orc.data_term(orc.Frame.from_scene(sc), g[f"{tag}_beta"], opt.sf_point_plane_weight)
        np.testing.assert_array_equal(taps.cpu().numpy()[match], t.taps)        # tap rows bit-exact


@pytest.mark.parametrize("state", ["f64", "f32"])
@pytest.mark.parametrize("name", GOLDENS_K)
def test_lm_matches_reference_golden(name, state):
    g, sc, opt = load_golden(name)
    K = sc.sf_knn_idx.shape[1]
    sf, inputs, new_data = torch_frame(sc)
    if state == "f32":
        from test_gpu_parity import _to_state32
        _to_state32(sf)
    lm = _solver(opt, K)
    beta = lm.LM(sf, inputs, new_data).cpu().numpy()
    recs = lm.last_records[0]
    assert all(r["status"] == 0 for r in recs)
    loss = np.array([r["loss"] for r in recs])
    np.testing.assert_allclose(loss, g["lm_loss"], rtol=1e-6 if state == "f64" else 1e-4, atol=1e-12)
    best = np.minimum.accumulate(np.concatenate([[1e10], g["lm_loss"]]))[:-1]
    decisive = np.abs(g["lm_loss"] - best) > 1e-6 * np.abs(best)
    acc = np.array([r["accepted"] for r in recs])
    np.testing.assert_array_equal(acc[decisive], g["lm_accepted"][decisive])
    np.testing.assert_allclose(beta, g["lm_beta"], rtol=0, atol=1e-7 if state == "f64" else 1e-4)
    assert [r["M_grad"] for r in recs][0] == len(g["b0_match"])


@pytest.mark.parametrize("state", ["f64", "f32"])
@pytest.mark.parametrize("name", GOLDENS_K)
def test_update_and_knn_feeder_match_reference_golden(name, state):
    import torch
    from super_amd import nodes
    from test_gpu_parity import _to_state32
    g, sc, opt = load_golden(name)
    K = sc.sf_knn_idx.shape[1]
    sf, _, _ = torch_frame(sc)
    if state == "f32":
        _to_state32(sf)
    sf.opt = _opt(opt, K)
    nodes.update(sf, torch.from_numpy(g["lm_beta"]).cuda())
    tol = 1e-12 if state == "f64" else 2e-7
    for mine, key in ((sf.points, "upd_points"), (sf.norms, "upd_norms"),
                      (sf.ED_nodes.points, "upd_ed_points"), (sf.ED_nodes.norms, "upd_ed_norms")):
        np.testing.assert_allclose(mine.cpu().numpy(), g[key], rtol=0, atol=tol)
    sf, _, _ = torch_frame(sc)
    if state == "f32":
        _to_state32(sf)
    sf.opt = _opt(opt, K)
    nodes.update_ed(sf)
    nodes.update_sfed_knn(sf)
    assert sf.knn_indices.shape[1] == K
    np.testing.assert_array_equal(sf.knn_indices.cpu().numpy(), g["knn_sf_idx"])   # bit-exact
    np.testing.assert_allclose(sf.knn_w.cpu().numpy(), g["knn_sf_w"], rtol=0, atol=1e-12 if state == "f64" else 1e-6)
    np.testing.assert_array_equal(sf.isStable.cpu().numpy(), g["knn_sf_stable"])


@pytest.mark.parametrize("K", [2, 3, 5, 8])
def test_other_neighbour_counts_against_the_oracle(K):
    """K = 2, 3, 5, 8 on a synthetic scene: three LM iterations against the (K-generic, golden-pinned) oracle."""
    from super_amd import synth
    sc = synth.make_scene(N=2500, J=60, H=60, W=80, seed=20 + K, src_border=5, tgt_border=3, n_neighbors=K)
    opt = orc.default_opt(num_optimize_iterations=3)
    sf, inputs, new_data = torch_frame(sc)
    lm = _solver(opt, K)
    beta = lm.LM(sf, inputs, new_data).cpu().numpy()
    trace = []
    want = orc.lm(orc.Frame.from_scene(sc), opt, trace=trace)
    np.testing.assert_allclose(beta, want, rtol=0, atol=1e-8)
    np.testing.assert_allclose([r["loss"] for r in lm.last_records[0]], [t["loss"] for t in trace], rtol=1e-8)
    assert [r["M_grad"] for r in lm.last_records[0]] == [t["M_grad"] for t in trace]


def test_k4_on_the_same_path_and_mixed_batches_are_refused():
    """K = 4 with ``slm_data_path = 1`` runs the very kernels K != 4 runs (instantiated for 4) and reproduces the
    reference golden; the frames of ONE batch must share their K (the per-surfel kernels are instantiated per K):
    a mixed batch is refused with SLM_ERR_UNSUPPORTED, an out-of-range K at the bind."""
    import torch
    from super_amd import _lib, synth
    from super_amd.engine import DeviceFrame, Engine
    g, sc, opt = load_golden("s60x80_j48")
    sf, inputs, new_data = torch_frame(sc)
    lm = _solver(opt, 4, slm_data_path=1)
    np.testing.assert_allclose(lm.LM(sf, inputs, new_data).cpu().numpy(), g["lm_beta"], rtol=0, atol=1e-7)
    dev = torch.device("cuda", 0)
    eng = Engine(dev, max_frames=2, num_iterations=2)
    eng.bind(0, DeviceFrame.from_scene(synth.make_scene(N=1500, J=48, H=60, W=80, seed=1, n_neighbors=4), dev))
    eng.bind(1, DeviceFrame.from_scene(synth.make_scene(N=1500, J=48, H=60, W=80, seed=2, n_neighbors=6), dev))
    assert eng.lib.slm_run(eng.h, 2, eng.stream) == _lib.SLM_ERR_UNSUPPORTED
    _lib.check(eng.lib.slm_run(eng.h, 1, eng.stream), "run of the K = 4 slot alone")
    eng.close()
    sc9 = synth.make_scene(N=1500, J=48, H=60, W=80, seed=3, n_neighbors=8)
    fr9 = DeviceFrame.from_scene(sc9, dev)
    eng = Engine(dev, max_frames=1, num_iterations=1)
    eng.bind(0, fr9)                                        # 8 is the largest supported value
    eng.close()
