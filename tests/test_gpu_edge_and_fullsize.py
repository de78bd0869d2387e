"""Edge cases and full-size (BASELINE C2) property tests of the HIP path, through the C ABI.
Needs an MI355X (-m gpu)."""
import ctypes as C

import numpy as np
import pytest

from helpers import ref_opt, torch_frame
from oracle import lm_oracle as orc

pytestmark = pytest.mark.gpu


def _engine(**kw):
    import torch
    from super_amd.engine import Engine
    return Engine(torch.device("cuda", 0), **kw)


def _dframe(sc):
    import torch
    from super_amd.engine import DeviceFrame
    return DeviceFrame.from_scene(sc, torch.device("cuda", 0))


# ------------------------------------------------------------------------------- edge cases
def test_no_surfel_matches_target_all_invalid():
    """Every target pixel invalid: empty match set, only ARAP/Rot act, beta stays identity."""
    from super_amd import synth
    sc = synth.make_scene(N=2000, J=48, H=60, W=80, seed=31, src_border=5, tgt_border=3)
    sc.valid[:] = False
    sc.index_map[:] = -1
    eng = _engine()
    eng.bind(0, _dframe(sc))
    eng.run(1)
    recs = eng.records(0)
    assert all(r["status"] == 0 and r["M_grad"] == 0 and r["M_loss"] == 0 for r in recs)
    beta = eng.beta(0).cpu().numpy()
    np.testing.assert_allclose(beta, np.tile([1.0, 0, 0, 0, 0, 0, 0], (sc.J, 1)), rtol=0, atol=1e-12)
    ob = orc.lm(orc.Frame.from_scene(sc), orc.default_opt())
    np.testing.assert_allclose(beta, ob, rtol=0, atol=1e-12)


def test_zero_surfels_and_single_iteration():
    from super_amd import synth
    sc = synth.make_scene(N=500, J=24, H=40, W=56, seed=32, src_border=4, tgt_border=2)
    for name in ("sf_points", "sf_norms", "sf_knn_idx", "sf_knn_w"):
        setattr(sc, name, getattr(sc, name)[:0].copy())
    eng = _engine(num_iterations=1)
    eng.bind(0, _dframe(sc))
    eng.run(1)
    recs = eng.records(0)
    assert len(recs) == 1 and recs[0]["status"] == 0 and recs[0]["M_grad"] == 0


def test_solver_failure_stops_like_the_reference():
    """Data term only, u0 = 0, nodes without any surfel -> singular JtJ -> 'Solver failed':
    the loop stops, beta is returned unchanged (super/LM.py:99-103)."""
    from super_amd import synth
    from super_amd import _lib
    sc = synth.make_scene(N=600, J=48, H=60, W=80, seed=33, src_border=5, tgt_border=3)
    keep = (sc.sf_knn_idx < 24).all(axis=1)          # surfels that only touch nodes 0..23
    for name in ("sf_points", "sf_norms", "sf_knn_idx", "sf_knn_w"):
        setattr(sc, name, np.ascontiguousarray(getattr(sc, name)[keep]))
    assert sc.N > 10
    eng = _engine(use_arap=False, use_rot=False, u0=0.0)
    eng.bind(0, _dframe(sc))
    eng.run(1)
    recs = eng.records(0)
    assert recs[0]["status"] == _lib.SLM_ITER_SOLVER_FAILED
    assert all(r["status"] == _lib.SLM_ITER_NOT_RUN for r in recs[1:])
    np.testing.assert_array_equal(eng.beta(0).cpu().numpy(), np.tile([1.0, 0, 0, 0, 0, 0, 0], (sc.J, 1)))
    # the Python mirror prints the reference's message
    import io, contextlib
    from super_amd.LM import LM_Solver
    o = ref_opt(orc.default_opt(mesh_arap=False, mesh_rot=False))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        LM_Solver(o).LM(*torch_frame(sc), u=0)
    assert "Solver failed: Ill-posed system!" in buf.getvalue()


def test_ragged_batch_and_k_ed_variants():
    """One batch with different N, J, image sizes and K_ED per slot equals the single solves."""
    from super_amd import synth
    scs = [synth.make_scene(N=2500, J=48, H=60, W=80, seed=41, src_border=5, tgt_border=3),
           synth.make_scene(N=900, J=24, H=40, W=56, seed=42, src_border=4, tgt_border=2, n_ed_neighbors=6),
           synth.make_scene(N=5000, J=108, H=120, W=160, seed=43, src_border=6, tgt_border=3, n_ed_neighbors=8)]
    singles = []
    for sc in scs:
        e = _engine()
        e.bind(0, _dframe(sc))
        e.run(1)
        singles.append(e.beta(0).cpu().numpy())
    eb = _engine(max_frames=3)
    for i, sc in enumerate(scs):
        eb.bind(i, _dframe(sc))
    eb.run(3)
    for i, sc in enumerate(scs):
        # same kernels, but the few f64 atomics (jtl, ARAP/Rot rows) sum in a different order
        # run to run; the late LM iterations (u ~ 1e-8) amplify that last-bit noise
        np.testing.assert_allclose(eb.beta(i).cpu().numpy(), singles[i], rtol=0, atol=1e-6)
        ob = orc.lm(orc.Frame.from_scene(sc), orc.default_opt())
        np.testing.assert_allclose(singles[i], ob, rtol=0, atol=1e-4)


def test_nan_target_rows_are_dropped_like_the_reference():
    from super_amd import synth
    sc = synth.make_scene(N=2000, J=48, H=60, W=80, seed=44, src_border=5, tgt_border=3)
    sc.tgt_points[::37] = np.nan
    sc.tgt_norms[5::53] = np.nan
    eng = _engine(num_iterations=2)
    eng.bind(0, _dframe(sc))
    eng.run(1)
    trace = []
    ob = orc.lm(orc.Frame.from_scene(sc), orc.default_opt(num_optimize_iterations=2), trace=trace)
    recs = eng.records(0)
    assert [r["M_grad"] for r in recs] == [t["M_grad"] for t in trace]
    assert recs[0]["M_grad"] < sc.N
    np.testing.assert_allclose(eng.beta(0).cpu().numpy(), ob, rtol=0, atol=1e-8)


def test_train_phase_always_accepts():
    from super_amd import synth
    sc = synth.make_scene(N=1500, J=48, H=60, W=80, seed=4, src_border=5, tgt_border=2, dphi=0.9)
    eng = _engine(phase_test=False)
    eng.bind(0, _dframe(sc))
    eng.run(1)
    recs = eng.records(0)
    assert all(r["accepted"] for r in recs)
    assert all(abs(r["u"] - 10.0) < 1e-12 for r in recs)       # u never changes in train phase
    ob = orc.lm(orc.Frame.from_scene(sc), orc.default_opt(phase="train"))
    np.testing.assert_allclose(eng.beta(0).cpu().numpy(), ob, rtol=0, atol=1e-6)


# ------------------------------------------------------------------- full size (BASELINE C2)
@pytest.fixture(scope="module")
def c2():
    from super_amd import synth
    return synth.make_scene(seed=0, **synth.WORKLOADS["C2"])


def test_c2_lm_properties(c2):
    """Size-independent properties at 200k surfels / 2k nodes: every iteration solves, accepted
    losses decrease, the damped normal equations are satisfied, u follows the accept history."""
    import torch
    from super_amd import _lib
    eng = _engine()
    eng.bind(0, _dframe(c2))
    eng.run(1)
    recs = eng.records(0)
    assert all(r["status"] == 0 for r in recs)
    assert recs[0]["M_grad"] > 0.95 * c2.N
    best, u = 1e10, 10.0
    for r in recs:
        assert abs(r["u"] - u) <= 1e-12 * u
        assert r["accepted"] == (r["loss"] < best)
        if r["accepted"]:
            best, u = r["loss"], u / 7.5
        else:
            u *= 7.5
    assert best < 0.05 * recs[0]["loss"] or best < recs[0]["loss"]
    # normal-equation residual of one damped solve at the final beta, multifrontal vs band
    P = 7 * c2.J
    sols = []
    for sp in (3, 1, 2):
        e2 = _engine(solver_path=sp)
        e2.bind(0, _dframe(c2))
        beta = eng.beta(0)
        _lib.check(e2.lib.slm_set_beta(e2.h, 0, beta.data_ptr(), e2.stream), "set_beta")
        d = torch.zeros(P, dtype=torch.float64, device="cuda")
        s = torch.zeros(1, dtype=torch.int32, device="cuda")
        _lib.check(e2.lib.slm_solve(e2.h, 0, 0.5, d.data_ptr(), s.data_ptr(), e2.stream), "solve")
        assert int(s.item()) == 0
        sols.append(d.cpu().numpy())
    np.testing.assert_allclose(sols[0], sols[1], rtol=0, atol=1e-8 * max(1.0, np.abs(sols[1]).max()))
    np.testing.assert_allclose(sols[0], sols[2], rtol=0, atol=1e-10 * max(1.0, np.abs(sols[0]).max()))   # [0]: path 3


def test_c2_assembly_paths_agree_and_match_oracle_terms(c2):
    """jtl of the tuple-sorted MFMA path == per-entry atomic path == NumPy oracle at C2; the
    match set is bit-exact against the oracle."""
    import torch
    from super_amd import _lib
    P = 7 * c2.J
    rng = np.random.default_rng(9)
    beta = np.tile([1.0, 0, 0, 0, 0, 0, 0], (c2.J, 1)) + np.concatenate(
        [rng.normal(0, 0.01, (c2.J, 4)), rng.normal(0, 0.002, (c2.J, 3))], axis=1)
    bt = torch.from_numpy(beta).cuda()
    out = []
    for dp in (0, 1, 2):
        e = _engine(data_path=dp, solver_path=1)
        e.bind(0, _dframe(c2))
        _lib.check(e.lib.slm_set_beta(e.h, 0, bt.data_ptr(), e.stream), "set_beta")
        jtl = torch.zeros(P, dtype=torch.float64, device="cuda")
        _lib.check(e.lib.slm_assemble(e.h, 0, None, jtl.data_ptr(), e.stream), "assemble")
        out.append(jtl.cpu().numpy())
        if dp == 0:
            m = torch.zeros(c2.N, dtype=torch.uint8, device="cuda")
            r = torch.zeros(c2.N, dtype=torch.float64, device="cuda")
            _lib.check(e.lib.slm_data_residuals(e.h, 0, r.data_ptr(), m.data_ptr(), None, e.stream), "resid")
            t = orc.data_term(orc.Frame.from_scene(c2), beta, 1.0)
            np.testing.assert_array_equal(np.nonzero(m.cpu().numpy())[0], t.match)
            np.testing.assert_allclose(r.cpu().numpy()[t.match], t.r, rtol=0, atol=1e-10)
    for k in (1, 2):
        np.testing.assert_allclose(out[0], out[k], rtol=0, atol=1e-9 * max(1.0, np.abs(out[k]).max()))


def test_c2_knn_sorted_and_update_identity(c2):
    import torch
    from super_amd import _lib
    lib = _lib.load()
    dev = torch.device("cuda", 0)
    q = torch.from_numpy(c2.sf_points).to(dev)
    n = torch.from_numpy(c2.ed_points).to(dev)
    idx = torch.empty((c2.N, 4), dtype=torch.int32, device=dev)
    dist = torch.empty((c2.N, 4), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.slm_knn(c2.N, c2.J, 4, 0, q.data_ptr(), n.data_ptr(), idx.data_ptr(), dist.data_ptr(), st), "knn")
    d = dist.cpu().numpy()
    assert (np.diff(d, axis=1) >= 0).all()                                  # ascending
    np.testing.assert_array_equal(idx.cpu().numpy(), c2.sf_knn_idx)        # == generator's KNN
    # Surfels.update with the identity warp leaves the points where they are (idempotence)
    p, nr = q.clone(), torch.from_numpy(c2.sf_norms).to(dev)
    g, gn = n.clone(), torch.from_numpy(c2.ed_norms).to(dev)
    beta = torch.zeros((c2.J, 7), dtype=torch.float64, device=dev)
    beta[:, 0] = 1.0
    w = torch.from_numpy(c2.sf_knn_w).to(dev)
    _lib.check(lib.slm_apply_update(c2.N, c2.J, 4, p.data_ptr(), nr.data_ptr(), idx.data_ptr(), w.data_ptr(),
                                    g.data_ptr(), gn.data_ptr(), beta.data_ptr(), st), "update")
    np.testing.assert_allclose(p.cpu().numpy(), c2.sf_points, rtol=0, atol=2e-7)
    np.testing.assert_allclose(g.cpu().numpy(), c2.ed_points, rtol=0, atol=0)
    np.testing.assert_allclose(np.linalg.norm(nr.cpu().numpy(), axis=1), 1.0, atol=1e-6)


def test_three_frame_tracking_loop_matches_oracle_pipeline():
    """The per-frame sequence SuPer.fusion runs (super/super.py:66-73 plus the KNN refresh):
    LM -> Surfels.update -> update_ed / update_sfed_knn -> next frame, three frames in a row,
    HIP mirrors vs the NumPy oracle doing the same sequence."""
    import torch
    from super_amd import nodes, synth
    from super_amd.LM import LM_Solver
    base = synth.make_scene(N=3000, J=48, H=60, W=80, seed=51, src_border=6, tgt_border=3, dphi=0.05)
    targets = [synth.make_scene(N=3000, J=48, H=60, W=80, seed=51, src_border=6, tgt_border=3,
                                dphi=0.05 * (k + 1)) for k in range(3)]
    opt = orc.default_opt(num_optimize_iterations=5)
    # ---- oracle pipeline: float64 state carried like the reference (nothing re-rounded; the HIP
    #      mirrors keep float64 state too)
    P, Nn = base.f64("sf_points"), base.f64("sf_norms")
    G, Gn, R = base.f64("ed_points"), base.f64("ed_norms"), base.f64("ed_radii")
    idx, w = base.sf_knn_idx.copy(), base.f64("sf_knn_w")
    eidx = base.ed_knn_idx.copy()
    # ---- HIP pipeline
    sf, inputs, _ = torch_frame(base)
    sf.opt = ref_opt(opt)
    lm = LM_Solver(sf.opt)
    for k, tg in enumerate(targets):
        fr = orc.Frame(sf_points=P, sf_knn_idx=idx, sf_knn_w=w, ed_points=G, ed_knn_idx=eidx,
                       tgt_points=tg.f64("tgt_points"), tgt_norms=tg.f64("tgt_norms"),
                       index_map=tg.index_map, valid=tg.valid, K=tg.K, H=tg.H, W=tg.W)
        beta_o = orc.lm(fr, opt)
        P, Nn, G, Gn = orc.apply_update(P, Nn, idx, w, G, Gn, beta_o)
        eidx, _, _ = orc.node_knn(G, R, 4)
        idx, w, _, _ = orc.surfel_knn(P, G, R, 4)

        _, _, new_data = torch_frame(tg)
        beta_h = lm.LM(sf, inputs, new_data)
        np.testing.assert_allclose(beta_h.cpu().numpy(), beta_o, rtol=0, atol=1e-4, err_msg=f"frame {k}")
        nodes.update(sf, beta_h)
        nodes.update_ed(sf)
        nodes.update_sfed_knn(sf)
        sf.ED_nodes.num = base.J
        np.testing.assert_allclose(sf.points.cpu().numpy(), P, rtol=0, atol=1e-8)
        np.testing.assert_allclose(sf.knn_w.cpu().numpy(), w, rtol=0, atol=1e-8)
        np.testing.assert_array_equal(sf.knn_indices.cpu().numpy(), idx)
        np.testing.assert_array_equal(sf.ED_nodes.knn_indices.cpu().numpy(), eidx)


def test_c4_semantic_graphfit_matches_oracle():
    """BASELINE configs[4] at full size (500k surfels / 4k nodes, 720x960, soft-seg point-plane +
    face + boundary morphing): one loss/gradient evaluation away from identity and 3 optimiser
    steps agree with the PyTorch-CPU oracle; the boundary pixels are bit-exact."""
    import torch
    from oracle import graphfit_oracle as gfo
    from super_amd import synth
    from super_amd.deform_mesh import GraphFit
    from helpers import GF_SEMANTIC_VARIANTS, torch_frame
    sc = synth.make_scene(seed=0, semantic=True, seg_smooth=11, **synth.WORKLOADS["C4"])
    opt = gfo.default_opt(**GF_SEMANTIC_VARIANTS["soft"])
    opt.deform_udpate_method, opt.num_classes, opt.num_optimize_iterations = "super_edg", 3, 3
    sf, inputs, new_data = torch_frame(sc)
    sf.ED_nodes.triangles = torch.from_numpy(sc.ed_triangles).cuda()
    sf.ED_nodes.triangles_areas = torch.from_numpy(sc.ed_triangle_areas).cuda().double()
    sf.seg = torch.from_numpy(sc.sf_seg).cuda()
    sf.seg_conf = torch.from_numpy(sc.sf_seg_conf).cuda()
    new_data.seg_conf = torch.from_numpy(sc.tgt_seg_conf).cuda()
    inputs[("seg_conf", 0)] = torch.from_numpy(sc.img_seg_conf).cuda()[None]
    inputs[("seg", 0)] = torch.from_numpy(sc.img_seg).cuda()[None, None]

    rng = np.random.default_rng(11)
    dv0 = np.tile(np.array([1.0, 0, 0, 0, 0, 0, 0]), (sc.J + 1, 1))
    dv0 += np.concatenate([rng.normal(0, 0.003, (sc.J + 1, 4)), rng.normal(0, 0.001, (sc.J + 1, 3))], axis=1)
    pb = gfo.Problem(sc)
    dvt = torch.from_numpy(dv0.copy()).requires_grad_(True)
    loss, terms = gfo.total_loss(pb, dvt, opt)
    gref, = torch.autograd.grad(loss, dvt)
    gref = gref.clone()
    gref[-1] /= sc.J
    gf = GraphFit(opt)
    t, matched, grad = gf.loss_and_grad(inputs, sf, new_data, torch.from_numpy(dv0).cuda())
    assert matched == terms["_matched"]
    assert gf.edge_counts == [len(e) for e in pb.edge_pts]
    for c in range(3):
        np.testing.assert_array_equal(gf.edge_points(c).cpu().numpy(), pb.edge_pts[c].numpy().astype(np.float32))
    assert gf.last_bn_morph_kept > 100
    for k, v in t.items():
        np.testing.assert_allclose(v, float(terms[k].detach()), rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(grad.cpu().numpy(), gref.numpy(), rtol=0,
                               atol=1e-9 * max(1.0, float(gref.abs().max())))
    # 3 SGD steps from identity: tolerance 1e-4 is the north_star bar; f64 everywhere gives far less
    dv = gf(inputs, sf, new_data, None).cpu().numpy()
    ref = gfo.graphfit(gfo.Problem(sc), opt)
    np.testing.assert_allclose(dv, ref, rtol=0, atol=1e-9)


@pytest.mark.parametrize("env", [{"SLM_ND_LEAF": "8"}, {"SLM_ND_LEAF": "96"}, {"SLM_COMPACT_MIN": "1"},
                                 {"SLM_COMPACT_MIN": "1000000"}, {"SLM_ND_LEAF": "40", "SLM_COMPACT_MIN": "4"}])
def test_solver_tuning_overrides_keep_the_result(env):
    """The plan / schedule knobs (leaf size of the dissection, compact-path threshold) change the elimination
    order and the launch sequence, not the solution: LM on a mid-size frame against the oracle under each
    override (read once per process, hence the subprocess)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, numpy as np\n"
        f"sys.path[:0] = [{root!r}, {os.path.join(root, 'python-super_amd')!r}, {os.path.join(root, 'tests')!r}]\n"
        "from helpers import ref_opt, torch_frame\n"
        "from oracle import lm_oracle as orc\n"
        "from super_amd import synth\n"
        "from super_amd.LM import LM_Solver\n"
        "sc = synth.make_scene(N=20000, J=300, H=240, W=320, seed=9)\n"
        "opt = orc.default_opt(num_optimize_iterations=3)\n"
        "lm = LM_Solver(ref_opt(opt), max_frames=2)\n"
        "b1 = lm.LM(*torch_frame(sc)).cpu().numpy()\n"
        "b2 = [b.cpu().numpy() for b in lm.LM_batch([torch_frame(sc), torch_frame(sc)])]\n"
        "want = orc.lm(orc.Frame.from_scene(sc), opt)\n"
        "print('ERR', float(np.abs(b1 - want).max()), float(np.abs(b2[0] - want).max()), float(np.abs(b2[1] - b2[0]).max()))\n")
    e = dict(os.environ)
    e.update(env)
    out = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    errs = [float(x) for x in out.stdout.strip().splitlines()[-1].split()[1:]]
    assert max(errs[:2]) < 1e-6 and errs[2] < 1e-12, (env, errs)   # (merged records add in arrival order)
