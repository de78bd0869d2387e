"""Multi-frame golden recorded from the REFERENCE itself (build container only): the model state is carried
in float64 from frame to frame exactly as ``SuPer.fusion`` does (``super/super.py:66-73``):

    beta = LM_Solver.LM(sf, inputs, sfdata)           super/LM.py:81-122
    Surfels.update(sf, beta)                          super/nodes.py:193-223
    Surfels.fuseInputData(sf, inputs, sfdata)         super/nodes.py:268-541
    Surfels.prepareStableIndexNSwapAllModel(...)      super/nodes.py:543-585

after ``update_ed`` / ``update_sfed_knn`` (``super/nodes.py:154-191``) initialised the skinning tables.  From
the first ``update`` on, surfel / node positions and ``knn_w`` are true float64 values (NOT float32
representable): this is the fixture that sees what a float32 shim would lose.  Recorded per frame: the state
handed to LM, the frame's ``sfdata``, the LM trace and beta, the match set and residuals at beta0, the state
after ``update``; the state after fusion + swap is the next frame's input.

    python tests/golden/make_golden_sequence.py      ->  tests/golden/seq_48x64.npz
"""
from __future__ import annotations

import logging
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "python-super_amd"))
sys.path.insert(0, HERE)

import ref_shim  # noqa: E402
import make_golden  # noqa: E402  (capture_lm / capture_terms)
from super_amd import synth  # noqa: E402

torch.set_num_threads(1)

N_FRAMES = 4
STATE = ("points", "norms", "colors", "radii", "confs", "time_stamp", "isStable", "knn_indices", "knn_w")


def target_frame(sc, t, rng):
    """sfdata of frame t: the analytic surface at phase phi + 0.15 t (float32 back-projection widened to
    float64 like the reference's depth_preprocessing), a few invalid pixels, random appearance fields."""
    H, W, K = sc.H, sc.W, sc.K
    vv, uu = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    noise = rng.normal(0.0, 1e-4, size=(H, W))
    tp, tn = synth._surface_points_normals(uu, vv, H, W, sc.meta["phi"] + 0.15 * t, K, noise)
    valid = np.zeros((H, W), bool)
    valid[3:H - 3, 3:W - 3] = True
    valid &= rng.uniform(size=(H, W)) >= 0.015
    index_map = -np.ones((H, W), np.int64)
    index_map[valid] = np.arange(int(valid.sum()))
    T = int(valid.sum())
    return dict(points=synth._f32(tp[valid]).astype(np.float64), norms=synth._f32(tn[valid]).astype(np.float64),
                colors=rng.uniform(0, 255, (T, 3)).astype(np.float32), radii=rng.uniform(0.002, 0.004, T),
                confs=rng.uniform(0.05, 1.0, T).astype(np.float32), valid=valid.reshape(-1).copy(),
                index_map=index_map)


def main():
    ref = ref_shim.install()
    sc = synth.make_scene(N=1100, J=30, H=48, W=64, seed=21, src_border=4, tgt_border=3, jitter=0.45)
    rng = np.random.default_rng(2100)
    t = lambda a: torch.from_numpy(np.array(a, copy=True))
    N = sc.N
    opt = ref_shim.ref_opt(height=sc.H, width=sc.W, th_dist=0.02, th_cosine_ang=0.4, th_time_steps=30,
                           disable_merging_new_surfels=False, disable_merging_exist_surfels=False,
                           disable_adding_new_surfels=False, disable_removing_unstable_surfels=False,
                           data="superv2", data_dir="")
    ed = ref_shim.Data(points=t(sc.f64("ed_points")), norms=t(sc.f64("ed_norms")), radii=t(sc.f64("ed_radii")),
                       num=sc.J, param_num=7 * sc.J)
    me = SimpleNamespace(opt=opt, hard_seg=False, evaluate_tracking=False, logger=logging.getLogger("seq"),
                         points=t(sc.f64("sf_points")), norms=t(sc.f64("sf_norms")),
                         colors=t(rng.uniform(0, 255, (N, 3)).astype(np.float32)), radii=t(rng.uniform(0.002, 0.004, N)),
                         confs=t(rng.uniform(0.2, 3.0, N).astype(np.float32)), time_stamp=torch.zeros(N),
                         isStable=torch.ones(N, dtype=torch.bool), ED_nodes=ed, projdata=torch.zeros(N, 2),
                         summary_writer=ref_shim._SummaryWriter(), render_img=lambda inputs: None,
                         viz=lambda inputs, sfdata: None, time=0)
    g = dict(H=sc.H, W=sc.W, K=sc.K, n_frames=N_FRAMES, J=sc.J,
             opt_th=np.array([opt.th_dist, opt.th_cosine_ang, opt.th_time_steps]),
             init_points=sc.f64("sf_points"), init_norms=sc.f64("sf_norms"), ed_radii=sc.f64("ed_radii"),
             init_ed_points=sc.f64("ed_points"), init_ed_norms=sc.f64("ed_norms"))
    # frame 0 of the reference: skinning tables from the KNN feeder (float64 weights from the start)
    ref.nodes.Surfels.update_ed(me)
    ref.nodes.Surfels.update_sfed_knn(me)
    g["ed_knn_idx"], g["ed_knn_w"] = ed.knn_indices.numpy().copy(), ed.knn_w.numpy().copy()
    g["init_isStable"] = me.isStable.numpy().copy()

    solver = ref.LM.LM_Solver(opt)
    for fi in range(1, N_FRAMES + 1):
        p = f"f{fi}_"
        fr = target_frame(sc, fi, rng)
        sfdata = ref_shim.Data(**{k: t(v) for k, v in fr.items()}, time=fi)
        inputs = {("color", 0): torch.zeros(1, 3, sc.H, sc.W), "K": torch.from_numpy(sc.K)[None], "ID": torch.tensor([fi]),
                  "time": fi, "filename": ["%06d" % fi]}
        for k, v in fr.items():
            g[p + "new_" + k] = v
        for k in STATE:                                   # the state LM reads (float64, carried)
            g[p + "in_" + k] = getattr(me, k).numpy().copy()
        g[p + "in_ed_points"], g[p + "in_ed_norms"] = ed.points.numpy().copy(), ed.norms.numpy().copy()
        # match set / residuals at the identity warp (pins the discrete decisions on float64 state)
        beta0 = torch.tensor([[1.0, 0, 0, 0, 0, 0, 0]], dtype=torch.float64).repeat(sc.J, 1)
        tm = make_golden.capture_terms(ref, solver, me, inputs, sfdata, beta0)
        g[p + "b0_match"], g[p + "b0_data_r"] = tm["match"].astype(np.int32), tm["data_r"]
        g[p + "b0_loss"], g[p + "b0_jtl"] = tm["loss"], tm["jtl"]
        g[p + "b0_u"], g[p + "b0_v"] = tm["u_"], tm["v_"]       # float projections: margins of the round / floor decisions
        lm = make_golden.capture_lm(ref, solver, me, inputs, sfdata)
        for k in ("lm_beta", "lm_loss", "lm_u", "lm_accepted"):
            g[p + k] = lm[k]
        ref.nodes.Surfels.update(me, torch.from_numpy(lm["lm_beta"]))
        g[p + "upd_points"], g[p + "upd_norms"] = me.points.numpy().copy(), me.norms.numpy().copy()
        g[p + "upd_ed_points"], g[p + "upd_ed_norms"] = ed.points.numpy().copy(), ed.norms.numpy().copy()
        ref.nodes.Surfels.fuseInputData(me, inputs, sfdata)
        g[p + "fuse_count"] = np.array(len(me.points))
        ref.nodes.Surfels.prepareStableIndexNSwapAllModel(me, inputs, sfdata)
        f32rep = float(np.mean(me.points.numpy().astype(np.float32).astype(np.float64) == me.points.numpy()))
        print(f"frame {fi}: T={len(fr['points'])} M(b0)={len(tm['match'])} loss {lm['lm_loss'][0]:.4e} -> {lm['lm_loss'][-1]:.4e} "
              f"accepted={lm['lm_accepted'].astype(int)} surfels {len(g[p + 'in_points'])} -> fuse {int(g[p + 'fuse_count'])} "
              f"-> swap {len(me.points)}; float32-representable coords after the frame: {100 * f32rep:.1f} %")
    for k in STATE:                                       # final state after the last fusion + swap
        g["final_" + k] = getattr(me, k).numpy().copy()
    g["final_ed_points"] = ed.points.numpy().copy()
    path = os.path.join(HERE, "seq_48x64.npz")
    np.savez_compressed(path, **g)
    print(path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
