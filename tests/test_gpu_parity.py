"""HIP path vs oracle / golden fixtures, through the C ABI.  Needs an MI355X (-m gpu)."""
import numpy as np
import pytest

from helpers import GOLDENS, coo_dense, load_golden, ref_opt, torch_frame
from oracle import lm_oracle as orc

pytestmark = pytest.mark.gpu

TOL_BETA = 1e-4      # north_star: node poses within 1e-4
TOL_F64 = 1e-9       # same f64 arithmetic, different reduction order


def _solver(opt, **kw):
    from super_amd.LM import LM_Solver
    return LM_Solver(ref_opt(opt), **kw)


@pytest.mark.parametrize("name", GOLDENS)
@pytest.mark.parametrize("tag", ["b0", "b1"])
def test_assemble_matches_reference_golden(name, tag):
    import torch
    g, sc, opt = load_golden(name)
    sf, inputs, new_data = torch_frame(sc)
    lm = _solver(opt)
    beta = torch.from_numpy(g[f"{tag}_beta"]).cuda()
    jtj, jtl = lm.prepareCostTerm(sf, inputs, new_data, beta, grad=True)
    loss = lm.prepareCostTerm(sf, inputs, new_data, beta, grad=False)
    np.testing.assert_allclose(jtl.cpu().numpy().reshape(-1), g[f"{tag}_jtl"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(float(loss), float(g[f"{tag}_loss"]), rtol=1e-8)
    P = 7 * sc.J
    fr = orc.Frame.from_scene(sc)
    JtJ, _, _ = orc.normal_equations(fr, g[f"{tag}_beta"], opt)
    np.testing.assert_allclose(jtj.cpu().numpy(), JtJ, rtol=0, atol=1e-7 * max(1.0, np.abs(JtJ).max()))
    if f"{tag}_jtj_nz_idx" in g.files:
        ref = coo_dense(g[f"{tag}_jtj_nz_idx"], g[f"{tag}_jtj_nz_val"], (P, P))
        np.testing.assert_allclose(jtj.cpu().numpy(), ref, rtol=0, atol=1e-7 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("name", GOLDENS)
def test_match_set_and_residuals_bit_exact_indices(name):
    import ctypes as C
    import torch
    from super_amd import _lib
    from super_amd.LM import _dev_ptr, _stream_ptr
    g, sc, opt = load_golden(name)
    if not opt.sf_point_plane:
        pytest.skip("no data term")
    sf, inputs, new_data = torch_frame(sc)
    lm = _solver(opt)
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
        np.testing.assert_allclose(r.cpu().numpy()[match], g[f"{tag}_data_r"], rtol=0, atol=TOL_F64)
        t = orc.data_term(orc.Frame.from_scene(sc), g[f"{tag}_beta"], opt.sf_point_plane_weight)
        np.testing.assert_array_equal(taps.cpu().numpy()[match], t.taps)        # tap rows bit-exact


@pytest.mark.parametrize("name", GOLDENS)
def test_lm_matches_reference_golden(name):
    g, sc, opt = load_golden(name)
    sf, inputs, new_data = torch_frame(sc)
    lm = _solver(opt)
    beta = lm.LM(sf, inputs, new_data).cpu().numpy()
    recs = lm.last_records[0]
    assert all(r["status"] == 0 for r in recs)
    loss = np.array([r["loss"] for r in recs])
    np.testing.assert_allclose(loss, g["lm_loss"], rtol=1e-6, atol=1e-12)
    # accept flags must agree wherever the decision is not a rounding-level tie
    best = np.minimum.accumulate(np.concatenate([[1e10], g["lm_loss"]]))[:-1]
    decisive = np.abs(g["lm_loss"] - best) > 1e-9 * np.abs(best)
    acc = np.array([r["accepted"] for r in recs])
    np.testing.assert_array_equal(acc[decisive], g["lm_accepted"][decisive])
    if decisive.all():
        np.testing.assert_allclose([r["u"] for r in recs], g["lm_u"], rtol=1e-12)
    np.testing.assert_allclose(beta, g["lm_beta"], rtol=0, atol=TOL_BETA)
    # residual norm within 1e-4 (north_star)
    assert abs(np.sqrt(loss.min()) - np.sqrt(g["lm_loss"].min())) < 1e-4


def _to_state32(sf):
    """the compact float32 state layout (BASELINE's fp32 configs): same values, float32 tensors"""
    for o, names in ((sf, ("points", "norms", "knn_w")), (sf.ED_nodes, ("points", "norms", "knn_w", "radii"))):
        for k in names:
            setattr(o, k, getattr(o, k).float())
    return sf


@pytest.mark.parametrize("state", ["f64", "f32"])
@pytest.mark.parametrize("name", ["s60x80_j48", "s120x160_j108"])
def test_update_matches_reference_golden(name, state):
    import torch
    from super_amd import nodes
    g, sc, opt = load_golden(name)
    sf, _, _ = torch_frame(sc)
    if state == "f32":
        _to_state32(sf)
    sf.opt = ref_opt(opt)
    nodes.update(sf, torch.from_numpy(g["lm_beta"]).cuda())
    assert sf.points.dtype == (torch.float64 if state == "f64" else torch.float32)
    tol = 1e-12 if state == "f64" else 2e-7      # float64 state like the reference / float32 storage
    for mine, key in ((sf.points, "upd_points"), (sf.norms, "upd_norms"),
                      (sf.ED_nodes.points, "upd_ed_points"), (sf.ED_nodes.norms, "upd_ed_norms")):
        np.testing.assert_allclose(mine.cpu().numpy(), g[key], rtol=0, atol=tol)


@pytest.mark.parametrize("state", ["f64", "f32"])
@pytest.mark.parametrize("name", ["s60x80_j48", "s120x160_j108"])
def test_knn_feeder_matches_reference_golden(name, state):
    from super_amd import nodes
    g, sc, opt = load_golden(name)
    sf, _, _ = torch_frame(sc)
    if state == "f32":
        _to_state32(sf)
    sf.opt = ref_opt(opt)
    nodes.update_ed(sf)
    nodes.update_sfed_knn(sf)
    np.testing.assert_array_equal(sf.knn_indices.cpu().numpy(), g["knn_sf_idx"])   # bit-exact
    np.testing.assert_array_equal(sf.ED_nodes.knn_indices.cpu().numpy(), g["knn_ed_idx"])
    tol = 1e-12 if state == "f64" else 1e-6
    np.testing.assert_allclose(sf.knn_w.cpu().numpy(), g["knn_sf_w"], rtol=0, atol=tol)
    np.testing.assert_allclose(sf.ED_nodes.knn_w.cpu().numpy(), g["knn_ed_w"], rtol=0, atol=tol)
    np.testing.assert_array_equal(sf.isStable.cpu().numpy(), g["knn_sf_stable"])


def test_knn_exact_ties_take_the_lowest_index():
    """pytorch3d's tie order is not pinned by the reference; the build's rule is lowest index
    first, in both feeder kernels, with exact float ties (a query equidistant from mirrored nodes)."""
    import torch
    from super_amd import nodes
    n = torch.tensor([[1.0, 0, 0], [-1.0, 0, 0], [0, 1.0, 0], [0, -1.0, 0], [0, 0, 1.0], [0, 0, -1.0], [2.0, 0, 0]],
                     dtype=torch.float64).cuda()
    q = torch.tensor([[0.0, 0, 0], [0.5, 0.5, 0.0]], dtype=torch.float64).cuda()
    for dt in (torch.float64, torch.float32):
        d, i = nodes.find_knn(q.to(dt), n.to(dt), k=4)
        np.testing.assert_array_equal(i.cpu().numpy(), [[0, 1, 2, 3], [0, 2, 4, 5]])
        np.testing.assert_allclose(d[0].cpu().numpy(), 1.0)
    from super_amd import _lib
    with pytest.raises(_lib.SuperLMError):          # fewer nodes than neighbours: refused, not index -1
        nodes.find_knn(q, n[:3], k=4)


def test_dense_solver_and_failure_status():
    import torch
    from super_amd.LM import LM_Solver
    rng = np.random.default_rng(0)
    for P in (7, 64, 200):
        A = rng.normal(size=(P, P))
        A = A @ A.T + P * np.eye(P)
        b = rng.normal(size=(P, 1))
        x = LM_Solver.Solver(torch.from_numpy(A).cuda(), torch.from_numpy(b).cuda())
        np.testing.assert_allclose(x.cpu().numpy(), np.linalg.solve(A, b), rtol=1e-9, atol=1e-11)
    A = -np.eye(5)
    with pytest.raises(RuntimeError):
        LM_Solver.Solver(torch.from_numpy(A).cuda(), torch.ones(5, 1, dtype=torch.float64).cuda())


def test_batch_of_frames_matches_single():
    from super_amd import synth
    scs = [synth.make_scene(N=3000, J=48, H=60, W=80, seed=s, src_border=5, tgt_border=3)
           for s in (11, 12, 13)]
    opt = orc.default_opt()
    lm1 = _solver(opt)
    singles = [lm1.LM(*torch_frame(sc)).cpu().numpy() for sc in scs]
    lmb = _solver(opt, max_frames=3)
    batch = lmb.LM_batch([torch_frame(sc) for sc in scs])
    for a, b in zip(singles, batch):
        np.testing.assert_allclose(b.cpu().numpy(), a, rtol=0, atol=1e-7)
    # and against the oracle
    ob = orc.lm(orc.Frame.from_scene(scs[0]), opt)
    np.testing.assert_allclose(singles[0], ob, rtol=0, atol=TOL_BETA)


def test_symbolic_plan_survives_a_changing_pair_list():
    """Surfels come and go between frames (fusion), so the coupled node-pair list changes a little every
    frame.  The solver keeps its symbolic plan for the union of the lists it has seen: frames solved one after
    the other in the same slot give the same warp as each solved by a fresh solver, and match the oracle."""
    import torch
    from super_amd import synth
    sc = synth.make_scene(N=20000, J=300, H=240, W=320, seed=21)
    opt = orc.default_opt(num_optimize_iterations=3)
    rng = np.random.default_rng(3)

    def variant(keep):
        sf, inputs, new_data = torch_frame(sc)
        if keep is not None:
            k = torch.from_numpy(keep).cuda()
            sf.points, sf.norms, sf.knn_indices, sf.knn_w = sf.points[k], sf.norms[k], sf.knn_indices[k], sf.knn_w[k]
            sf.isStable = sf.isStable[k]
        return sf, inputs, new_data

    keeps = [None] + [np.sort(rng.choice(sc.N, int(f * sc.N), replace=False)) for f in (0.9, 0.5, 0.95)] + [None]
    lm = _solver(opt)
    chained = [lm.LM(*variant(keep)).cpu().numpy() for keep in keeps]
    for keep, got in zip(keeps, chained):
        fresh = _solver(opt).LM(*variant(keep)).cpu().numpy()
        np.testing.assert_allclose(got, fresh, rtol=0, atol=1e-9)
    np.testing.assert_allclose(chained[0], chained[-1], rtol=0, atol=1e-12)    # same frame, first and last
    want = orc.lm(orc.Frame.from_scene(sc), opt)
    np.testing.assert_allclose(chained[0], want, rtol=0, atol=TOL_BETA)


@pytest.mark.parametrize("N,J", [(20000, 300), (50000, 512)])
def test_mid_size_vs_oracle(N, J):
    """C1-sized parity against the NumPy oracle (seconds on CPU)."""
    from super_amd import synth
    sc = synth.make_scene(N=N, J=J, H=240 if N <= 20000 else 480, W=320 if N <= 20000 else 640,
                          seed=5)
    opt = orc.default_opt(num_optimize_iterations=3)
    lm = _solver(opt)
    beta = lm.LM(*torch_frame(sc)).cpu().numpy()
    trace = []
    ob = orc.lm(orc.Frame.from_scene(sc), opt, trace=trace)
    np.testing.assert_allclose(beta, ob, rtol=0, atol=TOL_BETA)
    np.testing.assert_allclose([r["loss"] for r in lm.last_records[0]],
                               [t["loss"] for t in trace], rtol=1e-6)
    assert [r["M_loss"] for r in lm.last_records[0]] == [t["M_loss"] for t in trace]


def test_atomic_cross_check_path_agrees():
    """data_path=1 (per-entry f64 atomics), data_path=2 (one Gram per run in HBM) and the default
    tuple-sorted MFMA assembly with workgroup-merged records give the same normal equations."""
    import torch
    g, sc, opt = load_golden("s120x160_j108")
    sf, inputs, new_data = torch_frame(sc)
    beta = torch.from_numpy(g["b1_beta"]).cuda()
    outs = []
    for path in (0, 1, 2):
        o = ref_opt(opt)
        o.slm_data_path = path
        from super_amd.LM import LM_Solver
        lm = LM_Solver(o)
        jtj, jtl = lm.prepareCostTerm(sf, inputs, new_data, beta, grad=True)
        outs.append((jtj.cpu().numpy(), jtl.cpu().numpy()))
    scale = np.abs(outs[0][0]).max()
    for k in (1, 2):
        np.testing.assert_allclose(outs[0][0], outs[k][0], rtol=0, atol=1e-12 * scale)
        np.testing.assert_allclose(outs[0][1], outs[k][1], rtol=0, atol=1e-11)


@pytest.mark.parametrize("name", ["s60x80_j48", "s120x160_j108"])
def test_nd_multifrontal_solve_matches_band_and_numpy(name):
    """slm_solve through the nested-dissection multifrontal path, the band path, and a dense
    NumPy solve of the assembled system agree."""
    import torch
    from super_amd import _lib
    from super_amd.LM import LM_Solver, _dev_ptr, _stream_ptr
    g, sc, opt = load_golden(name)
    sf, inputs, new_data = torch_frame(sc)
    beta = torch.from_numpy(g["b1_beta"]).cuda()
    P = 7 * sc.J
    sols = {}
    for path in (3, 1, 2, 0):     # per-level launches, band, persistent task graph, default (picks by batch size)
        o = ref_opt(opt)
        o.slm_solver_path = path
        lm = LM_Solver(o)
        jtj, jtl = lm.prepareCostTerm(sf, inputs, new_data, beta, grad=True)   # binds + sets beta
        h = lm._handle()
        st = _stream_ptr(beta.device)
        delta = torch.zeros(P, dtype=torch.float64, device="cuda")
        status = torch.zeros(1, dtype=torch.int32, device="cuda")
        _lib.check(lm.lib.slm_solve(h, 0, 0.37, _dev_ptr(delta), _dev_ptr(status), st), "slm_solve")
        assert int(status.item()) == 0
        sols[path] = delta.cpu().numpy()
        A = jtj.cpu().numpy() + 0.37 * np.eye(P)
        ref = np.linalg.solve(A, jtl.cpu().numpy().reshape(-1))
        np.testing.assert_allclose(sols[path], ref, rtol=0, atol=1e-9 * max(1.0, np.abs(ref).max()))
    np.testing.assert_allclose(sols[3], sols[1], rtol=0, atol=1e-9)
    np.testing.assert_allclose(sols[3], sols[2], rtol=0, atol=1e-11)
    np.testing.assert_allclose(sols[0], sols[2], rtol=0, atol=1e-11)   # (the default data path is not bitwise reproducible)


def _run_emulated_ranks(name, world, extra_iterations=0):
    """One frame split over `world` ranks (one solver context each on this GPU; the all-reduce of
    the J^T J / J^T r pair blocks and of the loss partials and the broadcast of delta are emulated by
    combining the ranks' exchange buffers).  Returns (golden, opt, betas per rank, records per rank)."""
    import torch
    from super_amd import _lib
    from super_amd.LM import LM_Solver, _dev_ptr, _stream_ptr
    g, sc, opt = load_golden(name)
    sf, inputs, new_data = torch_frame(sc)
    o = ref_opt(opt)
    o.num_optimize_iterations = int(o.num_optimize_iterations) + extra_iterations
    ranks = [LM_Solver(o, rank=r, world=world, all_reduce=lambda t: None, broadcast=lambda t: None)
             for r in range(world)]
    hs = [lm._handle() for lm in ranks]
    bfs = [lm._bind(h, 0, sf, inputs, new_data) for lm, h in zip(ranks, hs)]
    dev = bfs[0].device
    st = _stream_ptr(dev)
    lib = ranks[0].lib

    def exchange(what, combine):
        bufs = [lm.exchange_buffer(h, 0, what, dev) for lm, h in zip(ranks, hs)]
        for h, b in zip(hs, bufs):
            _lib.check(lib.slm_lm_exchange_get(h, 0, what, _dev_ptr(b), st), "get")
        out = combine(bufs)
        for h in hs:
            _lib.check(lib.slm_lm_exchange_set(h, 0, what, _dev_ptr(out), st), "set")

    for _ in range(int(o.num_optimize_iterations)):
        for h in hs:
            _lib.check(lib.slm_lm_grad_local(h, 1, st), "grad_local")
        if o.sf_point_plane:
            exchange(_lib.SLM_X_PAIR_BLOCKS, lambda b: torch.stack(b).sum(0))
        for h in hs:
            _lib.check(lib.slm_lm_solve(h, 1, st), "solve")
        exchange(_lib.SLM_X_DELTA, lambda b: b[0].clone())
        for h in hs:
            _lib.check(lib.slm_lm_loss_local(h, 1, st), "loss_local")
        if o.sf_point_plane:
            exchange(_lib.SLM_X_DATA_LOSS, lambda b: torch.stack(b).sum(0))
        for h in hs:
            _lib.check(lib.slm_lm_accept(h, 1, st), "accept")
    betas = []
    for lm, h, bf in zip(ranks, hs, bfs):
        beta = torch.empty((bf.J, 7), dtype=torch.float64, device=dev)
        _lib.check(lib.slm_get_beta(h, 0, _dev_ptr(beta), st), "get_beta")
        betas.append(beta.cpu().numpy())
    recs = [lm.records(h, 0, st) for lm, h in zip(ranks, hs)]
    return g, o, betas, recs


@pytest.mark.parametrize("name,world", [("s60x80_j48", 2), ("s120x160_j108", 3), ("s60x80_j48_reject", 4)])
def test_surfel_sharded_lm_reproduces_the_single_gpu_solve(name, world):
    """Same iterations, same accept flags, same beta as the single-GPU solve and the reference's golden."""
    g, o, betas, recs = _run_emulated_ranks(name, world)
    for b in betas[1:]:
        np.testing.assert_array_equal(b, betas[0])                      # every rank: identical parameters
    for r in recs[1:]:
        assert [x["accepted"] for x in r] == [x["accepted"] for x in recs[0]]
        assert [x["loss"] for x in r] == [x["loss"] for x in recs[0]]
    np.testing.assert_allclose(betas[0], g["lm_beta"], rtol=0, atol=TOL_BETA)
    np.testing.assert_allclose([x["loss"] for x in recs[0]], g["lm_loss"], rtol=1e-6)
    assert [x["accepted"] for x in recs[0]] == [bool(a) for a in g["lm_accepted"]]
    assert recs[0][0]["M_grad"] == len(g["b0_match"])


@pytest.mark.parametrize("world", [2, 3])
def test_surfel_sharded_lm_keeps_the_matched_count_after_a_rejected_step(world):
    """after a rejected step the ranks send their Gram records AGAIN (the Jacobian pass is
    reused) -- and with them their OWN share of the matched-surfel count, not the all-reduced one of the iteration
    before (which would come back multiplied by the world size, compounding over consecutive rejects).  Runs past the
    fixture's reject and compares every record field with the unsharded solve."""
    from super_amd.LM import LM_Solver
    extra = 8
    g, o, betas, recs = _run_emulated_ranks("s60x80_j48_reject", world, extra_iterations=extra)
    _, sc, _ = load_golden("s60x80_j48_reject")
    lm = LM_Solver(o)
    want_beta = lm.LM(*torch_frame(sc)).cpu().numpy()
    want = lm.last_records[0]
    acc = [r["accepted"] for r in want]
    assert False in acc[:-1]                                   # a rejected iteration that is followed by another one
    for r in recs:
        assert [x["accepted"] for x in r] == acc
        assert [x["M_grad"] for x in r] == [x["M_grad"] for x in want]
        assert [x["M_loss"] for x in r] == [x["M_loss"] for x in want]
        assert [x["status"] for x in r] == [x["status"] for x in want]
    np.testing.assert_allclose(betas[0], want_beta, rtol=0, atol=1e-9)


def test_lm_is_bitwise_reproducible_with_the_per_run_slab():
    """No kernel of the LM path uses atomics on floating-point data any more except the LDS merge of
    the data-term records; with data_path=2 (one Gram per run) two runs give bit-identical results."""
    from super_amd.LM import LM_Solver
    g, sc, opt = load_golden("s120x160_j108")
    sf, inputs, new_data = torch_frame(sc)
    o = ref_opt(opt)
    o.slm_data_path = 2
    runs = []
    for _ in range(3):
        lm = LM_Solver(o)
        beta = lm.LM(sf, inputs, new_data).cpu().numpy()
        runs.append((beta, [r["loss"] for r in lm.last_records[0]]))
    for beta, losses in runs[1:]:
        np.testing.assert_array_equal(beta, runs[0][0])
        assert losses == runs[0][1]
    np.testing.assert_allclose(runs[0][0], g["lm_beta"], rtol=0, atol=TOL_BETA)


def test_rejected_iterations_reuse_the_jacobian_pass_bit_for_bit(monkeypatch):
    """After a rejected step beta is rolled back, so the next Jacobian pass would rebuild the same JtJ (reference
    super/LM.py:114-117, :96): the library keeps the Gram records of the slot instead.  On the run-to-run reproducible
    data path (data_path 2) the result must be BITWISE the one of recomputing (SLM_NO_REUSE=1), and the fixture must
    contain rejected iterations for the test to mean anything."""
    from helpers import load_golden
    import torch
    from super_amd.engine import DeviceFrame, Engine
    g, sc, opt = load_golden("s60x80_j48_reject")
    assert not g["lm_accepted"].all()
    n_it = int(opt.num_optimize_iterations) + 8      # the fixture's reject is its last iteration: run on past it
    out = []
    for no_reuse in ("0", "1"):
        monkeypatch.setenv("SLM_NO_REUSE", no_reuse)
        eng = Engine(torch.device("cuda", 0), data_path=2, num_iterations=n_it)
        eng.bind(0, DeviceFrame.from_scene(sc, torch.device("cuda", 0), state_f64=True))
        eng.run(1)
        recs = eng.records(0)
        out.append((eng.beta(0).cpu().numpy(), [(r["loss"], r["u"], r["accepted"], r["M_grad"], r["M_loss"]) for r in recs]))
        eng.close()
    acc = [r[2] for r in out[0][1]]
    assert acc[:len(g["lm_accepted"])] == [bool(a) for a in g["lm_accepted"]]
    assert False in acc[:-1]                         # a rejected iteration that is followed by another one
    assert out[0][1] == out[1][1]
    np.testing.assert_array_equal(out[0][0], out[1][0])
