"""Pin the oracles on a MULTI-FRAME fixture recorded from the reference with its float64 state carried
from frame to frame (tests/golden/make_golden_sequence.py): LM -> update -> fuseInputData -> swap, four
frames.  From the first update on the state is true float64 (not float32 representable), so this is the
fixture that would expose a float32 rounding of the model between frames.  CPU only."""
import os
from types import SimpleNamespace

import numpy as np
import pytest

from oracle import fusion_oracle as fuo
from oracle import lm_oracle as orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "seq_48x64.npz")
STATE = ("points", "norms", "colors", "radii", "confs", "time_stamp", "isStable", "knn_indices", "knn_w")


@pytest.fixture(scope="module")
def g():
    return np.load(GOLD)


def seq_frame(g, fi, state=None):
    """oracle Frame of frame fi; ``state`` overrides the recorded input state (free-running chain)."""
    p = f"f{fi}_"
    s = state or {k: g[p + "in_" + k] for k in STATE + ("ed_points", "ed_norms")}
    return orc.Frame(sf_points=s["points"], sf_knn_idx=s["knn_indices"], sf_knn_w=s["knn_w"], ed_points=s["ed_points"],
                     ed_knn_idx=g["ed_knn_idx"], tgt_points=g[p + "new_points"], tgt_norms=g[p + "new_norms"],
                     index_map=g[p + "new_index_map"], valid=g[p + "new_valid"], K=g["K"], H=int(g["H"]), W=int(g["W"]))


def fusion_opt(g):
    th = g["opt_th"]
    return fuo.default_opt(height=int(g["H"]), width=int(g["W"]), th_dist=float(th[0]), th_cosine_ang=float(th[1]),
                           th_time_steps=int(th[2]))


def fuse_and_swap(g, fi, s):
    """fuseInputData + prepareStableIndexNSwapAllModel of frame fi on state dict ``s`` (after update)."""
    p = f"f{fi}_"
    m = fuo.Model(s["points"], s["norms"], s["colors"], s["radii"], s["confs"], s["time_stamp"], s["isStable"],
                  s["knn_indices"], s["knn_w"], s["ed_points"], g["ed_radii"])
    new = SimpleNamespace(**{k: g[p + "new_" + k] for k in ("points", "norms", "colors", "radii", "confs", "valid")})
    opt = fusion_opt(g)
    fuo.fuse_input_data(m, opt, g["K"], new, fi)
    n_fused = len(m.points)
    fuo.swap_stable(m, opt, fi)
    out = {k: getattr(m, k) for k in STATE}
    out["ed_points"], out["ed_norms"] = s["ed_points"], s["ed_norms"]
    return out, n_fused


def test_state_is_true_float64(g):
    """the premise: after the first frame the carried state cannot be held in float32"""
    for fi in (2, 3, 4):
        p = g[f"f{fi}_in_points"]
        assert np.mean(p.astype(np.float32).astype(np.float64) == p) < 0.7
        w = g[f"f{fi}_in_knn_w"]
        assert np.mean(w.astype(np.float32).astype(np.float64) == w) < 0.1
    w1 = g["f1_in_knn_w"]          # weights from the float64 KNN feeder: not float32 values either
    assert np.mean(w1.astype(np.float32).astype(np.float64) == w1) < 0.1


def test_knn_feeder_initialises_the_tables(g):
    idx, w, stable, _ = orc.surfel_knn(g["init_points"], g["init_ed_points"], g["ed_radii"], 4)
    np.testing.assert_array_equal(idx, g["f1_in_knn_indices"])
    np.testing.assert_allclose(w, g["f1_in_knn_w"], rtol=0, atol=1e-13)
    np.testing.assert_array_equal(stable, g["init_isStable"])
    eidx, ew, _ = orc.node_knn(g["init_ed_points"], g["ed_radii"], 4)
    np.testing.assert_array_equal(eidx, g["ed_knn_idx"])
    np.testing.assert_allclose(ew, g["ed_knn_w"], rtol=0, atol=1e-13)


@pytest.mark.parametrize("fi", [1, 2, 3, 4])
def test_each_frame_on_the_reference_state(g, fi):
    """teacher-forced: every stage of frame fi from the state the reference handed it"""
    p = f"f{fi}_"
    fr = seq_frame(g, fi)
    opt = orc.default_opt()
    beta0 = np.tile(np.array([1.0, 0, 0, 0, 0, 0, 0]), (int(g["J"]), 1))
    t = orc.data_term(fr, beta0, 1.0)
    np.testing.assert_array_equal(t.match, g[p + "b0_match"])
    np.testing.assert_allclose(t.r, g[p + "b0_data_r"], rtol=0, atol=1e-10)
    trace = []
    beta = orc.lm(fr, opt, trace=trace)
    np.testing.assert_array_equal([x["accepted"] for x in trace], g[p + "lm_accepted"])
    np.testing.assert_allclose([x["loss"] for x in trace], g[p + "lm_loss"], rtol=1e-6, atol=1e-12)
    np.testing.assert_allclose(beta, g[p + "lm_beta"], rtol=0, atol=1e-6)
    pts, nrm, gp, gn = orc.apply_update(g[p + "in_points"], g[p + "in_norms"], g[p + "in_knn_indices"], g[p + "in_knn_w"],
                                        g[p + "in_ed_points"], g[p + "in_ed_norms"], g[p + "lm_beta"])
    np.testing.assert_allclose(pts, g[p + "upd_points"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(nrm, g[p + "upd_norms"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(gp, g[p + "upd_ed_points"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(gn, g[p + "upd_ed_norms"], rtol=0, atol=1e-13)
    s = {k: g[p + "in_" + k] for k in STATE}
    s.update(points=g[p + "upd_points"], norms=g[p + "upd_norms"], ed_points=g[p + "upd_ed_points"],
             ed_norms=g[p + "upd_ed_norms"])
    out, n_fused = fuse_and_swap(g, fi, s)
    assert n_fused == int(g[p + "fuse_count"])
    nxt = (lambda k: g[f"f{fi + 1}_in_{k}"]) if fi < int(g["n_frames"]) else (lambda k: g["final_" + k])
    assert len(out["points"]) == len(nxt("points"))
    for k in ("isStable", "knn_indices", "time_stamp"):
        np.testing.assert_array_equal(out[k], nxt(k))
    for k in ("points", "norms", "knn_w", "radii"):
        np.testing.assert_allclose(out[k], nxt(k), rtol=0, atol=1e-13)


def test_free_running_chain_stays_on_the_reference(g):
    """the oracle carries ITS OWN float64 state through all four frames (nothing re-rounded)"""
    s = {k: g["f1_in_" + k] for k in STATE + ("ed_points", "ed_norms")}
    opt = orc.default_opt()
    for fi in range(1, int(g["n_frames"]) + 1):
        p = f"f{fi}_"
        beta = orc.lm(seq_frame(g, fi, s), opt)
        np.testing.assert_allclose(beta, g[p + "lm_beta"], rtol=0, atol=1e-6)
        pts, nrm, gp, gn = orc.apply_update(s["points"], s["norms"], s["knn_indices"], s["knn_w"], s["ed_points"],
                                            s["ed_norms"], beta)
        s = dict(s, points=pts, norms=nrm, ed_points=gp, ed_norms=gn)
        s, n_fused = fuse_and_swap(g, fi, s)
        assert n_fused == int(g[p + "fuse_count"])
    assert len(s["points"]) == len(g["final_points"])
    np.testing.assert_array_equal(s["knn_indices"], g["final_knn_indices"])
    np.testing.assert_allclose(s["points"], g["final_points"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(s["knn_w"], g["final_knn_w"], rtol=0, atol=1e-7)
