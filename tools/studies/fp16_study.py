"""Evidence for the precision-mode decision of BASELINE.json configs[4] ("fp16 residuals + fp32 accum").

Emulates, on the CPU oracle (test infrastructure), what a reduced-precision storage mode of the LM hot path
would do to the result, and reports the node-pose error against the goldens recorded from the reference
(the parity bar of north_star: poses within 1e-4):

  tgt16    target point / normal tables stored as float16 (the 4-tap gather operands)
  jac16    data-term Jacobian rows and residuals rounded to float16, J^T J / J^T r accumulated in float32
  jac16s   as jac16 with the residuals scaled by 2^10 before the rounding (they are ~1e-3 m: float16 normal range
           starts at 6e-5) and un-scaled after the float32 accumulation
  all16    tgt16 + jac16s

    python tools/studies/fp16_study.py            (CPU, ~1 min)   ->  table on stdout (recorded in DESIGN.md)
"""
from __future__ import annotations

import os
import sys

import numpy as np
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "python-super_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import GOLDENS, load_golden  # noqa: E402
from oracle import lm_oracle as orc  # noqa: E402

f16 = lambda a: np.asarray(a, np.float64).astype(np.float16).astype(np.float64)


def normal_equations_16(scale):
    def ne(fr, beta, opt, dense=True):
        P = 7 * fr.J
        JtJ = np.zeros((P, P))
        jtl = np.zeros(P)
        M = 0
        for name, (rows, cols, vals, nrows, r, _) in orc.jacobian_coo(fr, beta, opt).items():
            if name == "data":
                M = nrows
                Jm = sp.coo_matrix((f16(vals).astype(np.float32), (rows, cols)), shape=(nrows, P)).tocsr()
                r16 = (f16(r * scale)).astype(np.float32)
                JtJ += (Jm.T @ Jm).astype(np.float32).toarray().astype(np.float64)      # float32 accumulation
                jtl -= (Jm.T @ r16).astype(np.float64) / scale
            else:
                Jm = sp.coo_matrix((vals, (rows, cols)), shape=(nrows, P)).tocsr()
                JtJ += (Jm.T @ Jm).toarray()
                jtl -= Jm.T @ r
        return JtJ, jtl, M
    return ne


def run(mode, fr, opt):
    keep = orc.normal_equations
    try:
        if mode in ("tgt16", "all16"):
            fr = orc.Frame(**{**fr.__dict__, "tgt_points": f16(fr.tgt_points), "tgt_norms": f16(fr.tgt_norms)})
        if mode == "jac16":
            orc.normal_equations = normal_equations_16(1.0)
        if mode in ("jac16s", "all16"):
            orc.normal_equations = normal_equations_16(1024.0)
        trace = []
        beta = orc.lm(fr, opt, trace=trace)
        return beta, [t.get("accepted") for t in trace]
    finally:
        orc.normal_equations = keep


def main():
    print(f"{'golden':24s} {'mode':7s} {'max |dq|':>10s} {'max |db| (m)':>12s} {'accept flips':>12s}")
    worst = {}
    for name in GOLDENS:
        g, sc, opt = load_golden(name)
        fr = orc.Frame.from_scene(sc) if hasattr(orc.Frame, "from_scene") else None
        if fr is None:
            fr = orc.Frame(sf_points=sc.f64("sf_points"), sf_knn_idx=sc.sf_knn_idx, sf_knn_w=sc.f64("sf_knn_w"),
                           ed_points=sc.f64("ed_points"), ed_knn_idx=sc.ed_knn_idx, tgt_points=sc.f64("tgt_points"),
                           tgt_norms=sc.f64("tgt_norms"), index_map=sc.index_map, valid=sc.valid, K=sc.K, H=sc.H, W=sc.W)
        ref, acc_ref = g["lm_beta"], list(g["lm_accepted"].astype(bool))
        for mode in ("f64", "tgt16", "jac16", "jac16s", "all16"):
            beta, acc = run(mode, fr, opt)
            dq = np.abs(beta[:, :4] - ref[:, :4]).max()
            db = np.abs(beta[:, 4:] - ref[:, 4:]).max()
            flips = sum(a != b for a, b in zip(acc, acc_ref))
            worst[mode] = max(worst.get(mode, 0.0), dq, db)
            print(f"{name:24s} {mode:7s} {dq:10.2e} {db:12.2e} {flips:12d}")
    print("worst pose error per mode:", {k: f"{v:.2e}" for k, v in worst.items()})


if __name__ == "__main__":
    main()
