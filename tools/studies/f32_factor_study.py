"""Gated study: would a MIXED-PRECISION solve hold the parity bars?

Mode under study: float32 storage of the assembled fronts + float32 Cholesky factor (half the bytes the multifrontal
solver moves, ~2x the MFMA rate, a shorter pivot chain), followed by k steps of float64 ITERATIVE REFINEMENT against
the float64 block-sparse JtJ (r = jtl - (JtJ + uI) x in float64, correction solved with the float32 factor).

Emulated on the CPU oracle (test infrastructure): every damped solve of the LM loop is replaced by
    A32 = float32(JtJ + uI);  L32 = chol(A32) in float32 (LAPACK spotrf);  x = L32-solve(jtl) in float32;
    k times:  r = jtl - A x (float64);  x += L32-solve(float32(r))
(a dense float32 factorisation stands in for the multifrontal one: same precision, same conditioning).
Gate, per fixture recorded from the reference (incl. the data-term-only problem and the four-frame sequence):
    * the accept / reject sequence is the float64 one,
    * the final beta is within 1e-7 of the float64 solver's (the bar the float64 path holds with margin; north_star's
      pose bar is 1e-4),
    * no float32 factorisation breaks down (non-positive pivot).

    python tools/studies/f32_factor_study.py            (CPU, ~1 min)  -> table on stdout (recorded in f32_factor_study.txt)
"""
from __future__ import annotations

import os
import sys

import numpy as np
from scipy.linalg import cho_factor, cho_solve

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "python-super_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import GOLDENS, load_golden  # noqa: E402
from oracle import lm_oracle as orc  # noqa: E402

STATS = {}


def make_solver(k_refine, prescale):
    def solve(JtJ, jtl, u):
        A = JtJ.copy()
        A[np.diag_indices_from(A)] += u
        # optional symmetric diagonal scaling (Jacobi) before the rounding to float32: D A D, D = diag(A)^-1/2
        d = 1.0 / np.sqrt(np.diag(A)) if prescale else np.ones(len(A))
        As = (A * d[:, None]) * d[None, :]
        A32 = As.astype(np.float32)
        try:
            c = cho_factor(A32, lower=True, check_finite=False)
        except np.linalg.LinAlgError:
            STATS["breakdowns"] = STATS.get("breakdowns", 0) + 1
            raise
        if not np.all(np.isfinite(c[0])):
            STATS["breakdowns"] = STATS.get("breakdowns", 0) + 1
            raise np.linalg.LinAlgError("float32 factor not finite")
        b = jtl * d
        x = cho_solve(c, b.astype(np.float32), check_finite=False).astype(np.float64)
        for _ in range(k_refine):
            r = b - As @ x
            x = x + cho_solve(c, r.astype(np.float32), check_finite=False).astype(np.float64)
        x = x * d
        res = np.abs(A @ x - jtl).max() / max(np.abs(jtl).max(), 1e-300)
        STATS["worst_residual"] = max(STATS.get("worst_residual", 0.0), res)
        STATS["worst_cond_u"] = min(STATS.get("worst_cond_u", 1e300), u)
        return x
    return solve


def run_lm(fr, opt, solver):
    keep = orc.solve_damped
    trace = []
    try:
        if solver is not None:
            orc.solve_damped = solver
        beta = orc.lm(fr, opt, trace=trace)
    finally:
        orc.solve_damped = keep
    return beta, trace


def sequence_frames():
    """The frames of the recorded four-frame sequence as independent LM problems (teacher-forced: every frame starts from
    the state the reference itself had, fixture keys f<k>_in_* / f<k>_new_*)."""
    g = np.load(os.path.join(ROOT, "tests", "golden", "seq_48x64.npz"))
    frames = []
    for k in range(1, int(g["n_frames"]) + 1):
        pre = f"f{k}_"
        if pre + "in_points" not in g.files:
            continue
        frames.append((k, orc.Frame(sf_points=g[pre + "in_points"], sf_knn_idx=g[pre + "in_knn_indices"].astype(np.int64),
                                    sf_knn_w=g[pre + "in_knn_w"], ed_points=g[pre + "in_ed_points"],
                                    ed_knn_idx=g["ed_knn_idx"].astype(np.int64), tgt_points=g[pre + "new_points"],
                                    tgt_norms=g[pre + "new_norms"], index_map=g[pre + "new_index_map"],
                                    valid=g[pre + "new_valid"].astype(bool), K=g["K"], H=int(g["H"]), W=int(g["W"])),
                       [bool(a) for a in g[pre + "lm_accepted"]]))
    return frames


def main():
    cases = []
    for name in GOLDENS:
        g, sc, opt = load_golden(name)
        cases.append((name, orc.Frame.from_scene(sc), opt))
    for k, fr, ref_acc in sequence_frames():
        cases.append((f"seq_48x64 f{k} " + "".join("T" if a else "F" for a in ref_acc), fr, orc.default_opt()))
    modes = [("f32", 0, False), ("f32+1", 1, False), ("f32+2", 2, False), ("f32s+1", 1, True), ("f32s+2", 2, True), ("f32s+3", 3, True)]
    print(f"{'fixture':26s} {'mode':8s} {'accept seq':>10s} {'max|dbeta|':>11s} {'worst solve resid':>18s} {'breakdowns':>10s} {'min u':>9s}  gate")
    all_ok = {m[0]: True for m in modes}
    for name, fr, opt in cases:
        ref_beta, ref_trace = run_lm(fr, opt, None)
        ref_acc = [t.get("accepted") for t in ref_trace]
        for label, k, scale in modes:
            STATS.clear()
            beta, trace = run_lm(fr, opt, make_solver(k, scale))
            acc = [t.get("accepted") for t in trace]
            same = acc == ref_acc
            err = float(np.abs(beta - ref_beta).max())
            ok = same and err < 1e-7 and STATS.get("breakdowns", 0) == 0
            all_ok[label] &= ok
            print(f"{name:26s} {label:8s} {'same' if same else 'DIFFERS':>10s} {err:11.2e} {STATS.get('worst_residual', float('nan')):18.2e} "
                  f"{STATS.get('breakdowns', 0):10d} {STATS.get('worst_cond_u', float('nan')):9.1e}  {'pass' if ok else 'FAIL'}")
    print()
    for label, ok in all_ok.items():
        print(f"gate for mode {label:7s}: {'PASS on every fixture' if ok else 'FAIL'}")


if __name__ == "__main__":
    main()
