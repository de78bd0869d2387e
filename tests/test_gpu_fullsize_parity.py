"""Full-size parity of the LM path against INDEPENDENT references, through the C ABI.

* BASELINE configs[2] (C2: 200 k surfels / 2 k nodes, point-to-plane + ARAP + Rot): three LM iterations of every
  numeric form of the solver -- task graph (one frame per launch), per-level launches, hybrid (a three-frame batch)
  -- against the NumPy oracle's dense float64 Cholesky (``oracle.lm_oracle.lm``; reference ``super/LM.py:95-117``,
  ``super/loss.py:222-290``): beta <= 1e-4 (north_star), loss trace 1e-6 relative, match counts and accept flags equal.
* BASELINE configs[1] AS WRITTEN (C1: 50 k / 512, point-to-plane ONLY, 10 iterations) against the oracle: the
  ill-conditioned case (nothing but the damping on the diagonal of unobserved directions).
* The damped normal equations themselves at C2 and C4: ``(JtJ + uI) delta = jtl`` with JtJ / jtl from
  ``slm_assemble`` (the block-banded assembly: other kernels than the fronts') -- the residual of every solver form's
  delta (float64 matrix-vector product in torch, on the device), and at C2 the same sparse system solved by SciPy's
  SuperLU on the host.

Needs an MI355X (-m gpu).  The oracle costs ~4 s per C2 iteration on the GPU box's host."""
import ctypes as C

import numpy as np
import pytest

from oracle import lm_oracle as orc

pytestmark = pytest.mark.gpu

_ORACLE = {}


def _engine(**kw):
    import torch
    from super_amd.engine import Engine
    return Engine(torch.device("cuda", 0), **kw)


def _dframe(sc):
    import torch
    from super_amd.engine import DeviceFrame
    return DeviceFrame.from_scene(sc, torch.device("cuda", 0))


def _scene(workload, seed):
    from super_amd import synth
    key = ("scene", workload, seed)
    if key not in _ORACLE:
        _ORACLE[key] = synth.make_scene(seed=seed, **synth.WORKLOADS[workload])
    return _ORACLE[key]


def _oracle(workload, seed, n_it, **opt_kw):
    key = (workload, seed, n_it, tuple(sorted(opt_kw.items())))
    if key not in _ORACLE:
        trace = []
        beta = orc.lm(orc.Frame.from_scene(_scene(workload, seed)),
                      orc.default_opt(num_optimize_iterations=n_it, **opt_kw), trace=trace)
        _ORACLE[key] = (beta, trace)
    return _ORACLE[key]


def _check_against_oracle(recs, beta, want_beta, trace, tag):
    assert [r["status"] for r in recs] == [0] * len(trace), tag
    assert [r["M_grad"] for r in recs] == [t["M_grad"] for t in trace], tag       # match sets: same size every pass
    assert [r["M_loss"] for r in recs] == [t["M_loss"] for t in trace], tag
    np.testing.assert_allclose([r["loss"] for r in recs], [t["loss"] for t in trace], rtol=1e-6, atol=1e-12, err_msg=tag)
    np.testing.assert_allclose([r["u"] for r in recs], [t["u"] for t in trace], rtol=1e-12, err_msg=tag)
    assert [r["accepted"] for r in recs] == [t["accepted"] for t in trace], tag
    err = float(np.abs(beta - want_beta).max())
    print(f"[{tag}] max|beta_hip - beta_oracle| = {err:.3e}")
    assert err < 1e-4, (tag, err)                                                  # north_star's bar
    return err


# ------------------------------------------------------------------ configs[2]: C2, all three terms
@pytest.mark.parametrize("solver_path,form", [(0, 1), (3, 0), (4, 2)])
def test_c2_three_iterations_match_the_oracle_one_frame_per_launch(solver_path, form):
    sc = _scene("C2", 0)
    want, trace = _oracle("C2", 0, 3)
    eng = _engine(num_iterations=3, solver_path=solver_path)
    eng.bind(0, _dframe(sc))
    eng.run(1)
    assert eng.lib.slm_debug_last_solver_form(eng.h) == form
    err = _check_against_oracle(eng.records(0), eng.beta(0).cpu().numpy(), want, trace, f"C2 B=1 path {solver_path}")
    assert err < 1e-7          # observed ~1e-11: anything near the bar would be a defect


@pytest.mark.parametrize("solver_path,form", [(4, 2), (0, 1)])
def test_c2_three_frame_batch_matches_the_oracle(solver_path, form):
    """A batch of three C2 frames: the hybrid form (what bench.py's 8 frames take under solver_path 0) and what
    solver_path 0 picks for three -- one task graph over the batch (frames x nodes <= SLM_DAG_MAX_NODES)."""
    seeds = (0, 1, 2)
    eng = _engine(num_iterations=3, max_frames=3, solver_path=solver_path)
    eng.bind_batch([_dframe(_scene("C2", s)) for s in seeds])
    eng.run(3)
    assert eng.lib.slm_debug_last_solver_form(eng.h) == form
    for i, s in enumerate(seeds):
        want, trace = _oracle("C2", s, 3)
        err = _check_against_oracle(eng.records(i), eng.beta(i).cpu().numpy(), want, trace, f"C2 B=3 slot {i}")
        assert err < 1e-7


# ------------------------------------------------------------------ configs[1] as written: C1, data term only, 10 it
@pytest.mark.parametrize("solver_path", [0, 3])
def test_c1_point_to_plane_only_ten_iterations_match_the_oracle(solver_path):
    sc = _scene("C1", 0)
    want, trace = _oracle("C1", 0, 10, mesh_arap=False, mesh_rot=False)
    assert len(trace) == 10
    eng = _engine(num_iterations=10, use_arap=False, use_rot=False, solver_path=solver_path)
    eng.bind(0, _dframe(sc))
    eng.run(1)
    recs = eng.records(0)
    # accept decisions are compared where they are decisive (a loss within 1e-9 of the best so far is a tie that
    # float64 summation order may break either way)
    best = 1e10
    for r, t in zip(recs, trace):
        if abs(t["loss"] - best) > 1e-9 * max(abs(best), 1e-30):
            assert r["accepted"] == t["accepted"]
        if t["accepted"]:
            best = t["loss"]
    assert [r["status"] for r in recs] == [0] * 10
    assert [r["M_grad"] for r in recs] == [t["M_grad"] for t in trace]
    np.testing.assert_allclose([r["loss"] for r in recs], [t["loss"] for t in trace], rtol=1e-6, atol=1e-12)
    err = float(np.abs(eng.beta(0).cpu().numpy() - want).max())
    print(f"[C1 data-only 10 it, path {solver_path}] max|beta_hip - beta_oracle| = {err:.3e}")
    assert err < 1e-4


# ------------------------------------------------------------------ the normal equations, checked independently
def _assembled_system(workload, seed, beta_t):
    """Dense JtJ (P,P) and jtl (P) on the device at `beta_t` from slm_assemble (block-banded assembly kernels)."""
    import torch
    from super_amd import _lib
    sc = _scene(workload, seed)
    P = 7 * sc.J
    e = _engine(solver_path=1)
    e.bind(0, _dframe(sc))
    _lib.check(e.lib.slm_set_beta(e.h, 0, beta_t.data_ptr(), e.stream), "set_beta")
    A = torch.empty((P, P), dtype=torch.float64, device="cuda")
    b = torch.empty(P, dtype=torch.float64, device="cuda")
    _lib.check(e.lib.slm_assemble(e.h, 0, A.data_ptr(), b.data_ptr(), e.stream), "assemble")
    torch.cuda.synchronize()
    e.close()
    return A, b


def _perturbed_beta(J, seed):
    import torch
    rng = np.random.default_rng(seed)
    beta = np.tile([1.0, 0, 0, 0, 0, 0, 0], (J, 1)) + np.concatenate(
        [rng.normal(0, 0.01, (J, 4)), rng.normal(0, 0.002, (J, 3))], axis=1)
    return torch.from_numpy(beta).cuda()


@pytest.mark.parametrize("workload", ["C2", "C4"])
def test_every_solver_form_satisfies_the_damped_normal_equations(workload):
    import torch
    from super_amd import _lib
    sc = _scene(workload, 0)
    P = 7 * sc.J
    bt = _perturbed_beta(sc.J, 17)
    A, b = _assembled_system(workload, 0, bt)
    assert torch.equal(A, A.T)
    bn = float(b.abs().max())
    deltas = {}
    for u in (10.0, 10.0 / 7.5 ** 5):                 # the first iteration's damping and a late one (4e-4)
        for sp in (2, 3, 4, 1):                       # task graph, per-level, hybrid, block-banded
            e = _engine(solver_path=sp)
            e.bind(0, _dframe(sc))
            _lib.check(e.lib.slm_set_beta(e.h, 0, bt.data_ptr(), e.stream), "set_beta")
            d = torch.zeros(P, dtype=torch.float64, device="cuda")
            s = torch.zeros(1, dtype=torch.int32, device="cuda")
            _lib.check(e.lib.slm_solve(e.h, 0, u, d.data_ptr(), s.data_ptr(), e.stream), "solve")
            assert int(s.item()) == 0
            r = A @ d + u * d - b
            rel = float(r.abs().max()) / bn
            print(f"[{workload} u={u:.3g} path {sp}] |A d - jtl|_inf / |jtl|_inf = {rel:.2e}")
            assert rel <= 1e-9, (workload, u, sp, rel)
            deltas[(u, sp)] = d.cpu().numpy()
            e.close()
    if workload == "C2":
        # the same sparse system through SciPy's SuperLU on the host: an independent SOLVER
        import scipy.sparse as sps
        import scipy.sparse.linalg as spla
        Asp = A.to_sparse().coalesce()
        ij, v = Asp.indices().cpu().numpy(), Asp.values().cpu().numpy()
        M = sps.csc_matrix((v, (ij[0], ij[1])), shape=(P, P))
        assert M.nnz < 0.01 * P * P                    # block-sparse: ~15.5 k 7x7 blocks
        for u in (10.0, 10.0 / 7.5 ** 5):
            x = spla.splu((M + u * sps.identity(P, format="csc")).tocsc()).solve(b.cpu().numpy())
            for sp in (2, 3, 4):
                err = float(np.abs(deltas[(u, sp)] - x).max()) / max(1.0, float(np.abs(x).max()))
                print(f"[C2 u={u:.3g} path {sp}] |delta - delta_SuperLU|_inf (rel) = {err:.2e}")
                assert err <= 1e-8
