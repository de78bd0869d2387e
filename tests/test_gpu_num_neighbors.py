"""``--num_neighbors`` other than 4: a documented tunable of the reference (``README.md:175``,
``options.py:49``; its code is K-generic: ``super/loss.py:213-220``, ``super/utils.py:30-36``, ``super/nodes.py:170-191``).
Round 6: K != 4 is a first-class path -- the K-generic pair plan (``prep_pairs``: the coupled-pair list and every surfel's
pair indices from one sort), the data term through per-pair records (``k_data_grad_pairs<K>``: lane = entry, run-length
accumulation in neighbour-set order) and the SAME nested-dissection multifrontal solver K = 4 runs (task graph / per-level
launches / hybrid); the surfel-sharded mode accepts it.  ``slm_data_path = 1`` / ``slm_solver_path = 1`` keep the round-5
compatibility path (per-entry atomics into the band + block-banded solve) as a cross-check.  ``Surfels.update`` and the KNN
feeder take K at run time.  Pinned by ``tests/golden/s60x80_j48_k6.npz``, recorded from the reference at ``num_neighbors = 6``
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


@pytest.mark.parametrize("K", [3, 6, 8])
def test_every_solver_form_runs_the_k_generic_pair_path(K):
    """Round 6: num_neighbors != 4 takes the multifrontal solver in every form (the plan is built from the coupling graph,
    whatever K produced it): one frame as a task graph, a batch of 8 as one task graph / per-level launches / hybrid, and
    the round-5 compatibility path (per-entry atomics + block-banded solve) -- the same four iterations, against the oracle."""
    import torch
    from super_amd import synth
    from super_amd.engine import DeviceFrame, Engine
    dev = torch.device("cuda", 0)
    scs = [synth.make_scene(N=12000, J=300, H=240, W=320, seed=300 + 10 * K + k, src_border=8, tgt_border=4, dphi=0.1 + 0.02 * k,
                            n_neighbors=K) for k in range(8)]
    opt = orc.default_opt(num_optimize_iterations=4)
    want = [orc.lm(orc.Frame.from_scene(sc), opt) for sc in scs[:2]]
    frames = [DeviceFrame.from_scene(sc, dev) for sc in scs]
    out = {}
    for name, kw, form in (("graph1", dict(max_frames=1), 1), ("graph8", dict(max_frames=8, solver_path=2), 1),
                           ("levels8", dict(max_frames=8, solver_path=3), 0), ("hybrid8", dict(max_frames=8, solver_path=4), 2),
                           ("band8", dict(max_frames=8, solver_path=1), -1)):
        e = Engine(dev, num_iterations=4, **kw)
        n = kw["max_frames"]
        e.bind_batch(frames[:n]) if n > 1 else e.bind(0, frames[0])
        e.run(n)
        assert e.lib.slm_debug_last_solver_form(e.h) == form, (name, e.lib.slm_debug_last_solver_form(e.h))
        out[name] = [(e.beta(i).cpu().numpy(), e.records(i)) for i in range(min(n, 2))]
        e.close()
    for name, res in out.items():
        for i, (b, r) in enumerate(res):
            assert all(x["status"] == 0 for x in r), name
            np.testing.assert_allclose(b, want[i], rtol=0, atol=1e-7, err_msg=name)
            assert [x["accepted"] for x in r] == [x["accepted"] for x in out["band8"][i][1]], name
            np.testing.assert_allclose([x["loss"] for x in r], [x["loss"] for x in out["band8"][i][1]], rtol=1e-9, err_msg=name)


@pytest.mark.parametrize("world", [2, 3])
def test_surfel_sharded_lm_accepts_k6(world):
    """The surfel-sharded LM mode at num_neighbors = 6 (it refused K != 4 through round 5): each emulated rank adds its
    share of the neighbour-set-ordered surfel list into its pair records, the records are all-reduced, every rank solves --
    identical parameters on every rank, the reference's golden trace."""
    from test_gpu_parity import TOL_BETA, _run_emulated_ranks
    g, o, betas, recs = _run_emulated_ranks("s60x80_j48_k6", world)
    for b in betas[1:]:
        np.testing.assert_array_equal(b, betas[0])
    for r in recs[1:]:
        assert [x["accepted"] for x in r] == [x["accepted"] for x in recs[0]]
        assert [x["loss"] for x in r] == [x["loss"] for x in recs[0]]
    np.testing.assert_allclose(betas[0], g["lm_beta"], rtol=0, atol=TOL_BETA)
    np.testing.assert_allclose([x["loss"] for x in recs[0]], g["lm_loss"], rtol=1e-6)
    best = np.minimum.accumulate(np.concatenate([[1e10], g["lm_loss"]]))[:-1]
    decisive = np.abs(g["lm_loss"] - best) > 1e-6 * np.abs(best)
    acc = np.array([x["accepted"] for x in recs[0]])
    np.testing.assert_array_equal(acc[decisive], g["lm_accepted"][decisive])
    assert recs[0][0]["M_grad"] == len(g["b0_match"])
    assert all(x["M_grad"] == y["M_grad"] for r in recs for x, y in zip(r, recs[0]))
