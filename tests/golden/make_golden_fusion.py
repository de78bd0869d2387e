"""Golden vectors for SURVEY.md 8(f) row f1: run the REFERENCE's ``Surfels.fuseInputData`` and
``Surfels.prepareStableIndexNSwapAllModel`` (``super/nodes.py:268-541,543-585``, unmodified, through
``ref_shim``) on a seeded synthetic surfel model + new frame and record inputs and outputs per
option variant.

    python tests/golden/make_golden_fusion.py        ->  tests/golden/fu_48x64.npz, fu_48x64_k6.npz (num_neighbors = 6, round 6)
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
from super_amd import synth  # noqa: E402

torch.set_num_threads(1)

VARIANTS = {
    "default": dict(),
    "tight": dict(th_dist=0.008, th_cosine_ang=0.9),          # many new pixels fail the merge test
    "nomerge_new": dict(disable_merging_new_surfels=True, _store_swap=False),
    "nomerge_exist": dict(disable_merging_exist_surfels=True, th_dist=0.008, _store_swap=False),
    "noadd": dict(disable_adding_new_surfels=True, th_dist=0.008, _store_swap=False),
    "keepall": dict(disable_removing_unstable_surfels=True, th_dist=0.02),
    # tracked evaluation points: some ids sit on surfels that get absorbed / deleted / go stale
    "track": dict(th_dist=0.02, _track=True),
    # Semantic-SuPer: segmentation fields fused and carried, Jensen-Shannon skinning weights
    "sem": dict(method="semantic-super", num_classes=3, th_dist=0.02, _seg=True),
    # hard_seg: only equal classes merge, new surfels take neighbours of their own class
    "hard": dict(method="semantic-super", num_classes=3, th_dist=0.02, _seg=True, _hard=True),
    # opt.data == "superv1" with segmentation fields: class test in the merge, plain weights
    "v1seg": dict(data="superv1", num_classes=3, th_dist=0.02, _seg=True),
}

STATE = ("points", "norms", "colors", "radii", "confs", "time_stamp", "isStable", "knn_indices", "knn_w", "projdata")
SEG_STATE = ("seg", "seg_conf", "dist2edge")


def make_inputs(seed=11, num_neighbors=4):
    """Surfel model (with several surfels on many pixels and a few unstable / stale ones) + new frame."""
    sc = synth.make_scene(N=1500, J=30, H=48, W=64, seed=seed, src_border=4, tgt_border=5, tgt_holes=0.02,
                          jitter=0.49, semantic=True, num_classes=3, n_neighbors=num_neighbors)
    rng = np.random.default_rng(seed)
    # surfels are sampled one per pixel: add perturbed copies so that many pixels carry 2-4 layers
    P0, N0, I0, W0 = sc.f64("sf_points"), sc.f64("sf_norms"), sc.sf_knn_idx, sc.f64("sf_knn_w")
    dup = np.concatenate([rng.choice(sc.N, 500, replace=False), rng.choice(sc.N, 200, replace=False),
                          rng.choice(sc.N, 80, replace=False)])
    scale = np.concatenate([np.full(500, 2e-3), np.full(200, 6e-3), np.full(80, 1.5e-2)])[:, None]
    Pd = P0[dup] + rng.normal(0, 1, (len(dup), 3)) * scale * np.array([0.2, 0.2, 1.0])
    Nd = N0[dup] + rng.normal(0, 0.15, (len(dup), 3))
    Nd /= np.linalg.norm(Nd, axis=1, keepdims=True)
    sf_points, sf_norms = np.concatenate([P0, Pd]), np.concatenate([N0, Nd])
    sf_knn_idx, sf_knn_w = np.concatenate([I0, I0[dup]]), np.concatenate([W0, W0[dup]])
    N, T = len(sf_points), sc.T
    base = dict(H=sc.H, W=sc.W, K=sc.K, num_neighbors=num_neighbors,
                sf_points=sf_points, sf_norms=sf_norms,
                sf_colors=rng.uniform(0, 255, (N, 3)).astype(np.float32),
                sf_radii=rng.uniform(0.002, 0.004, N),
                sf_confs=rng.uniform(0.2, 3.0, N).astype(np.float32),
                sf_time_stamp=(40.0 - rng.integers(0, 45, N)).astype(np.float32),   # some older than th_time_steps
                sf_isStable=rng.uniform(size=N) > 0.05,
                sf_knn_idx=sf_knn_idx, sf_knn_w=sf_knn_w,
                ed_points=sc.f64("ed_points"), ed_radii=sc.f64("ed_radii"),
                new_points=sc.f64("tgt_points"), new_norms=sc.f64("tgt_norms"),
                new_colors=rng.uniform(0, 255, (T, 3)).astype(np.float32),
                new_radii=rng.uniform(0.002, 0.004, T),
                new_confs=rng.uniform(0.05, 1.0, T).astype(np.float32),
                new_valid=sc.valid, new_index_map=sc.index_map, time=41)
    # 20 tracked ids: duplicated surfels (likely absorbed), their originals, stale ones, unassigned (-1)
    n0 = sc.N
    stale = np.nonzero(base["sf_time_stamp"] < 11)[0]
    base["track_id"] = np.concatenate([np.arange(n0, n0 + 8), dup[:4], stale[:4], rng.choice(n0, 2), [-1, -1]]).astype(np.int64)
    # segmentation fields (own generator: drawn after everything above so the other inputs keep their values)
    rs = np.random.default_rng(seed + 1000)
    C = sc.num_classes

    def smooth_conf(conf, noise):
        c = np.asarray(conf, np.float64) + rs.uniform(0, noise, conf.shape)
        return c / c.sum(1, keepdims=True)

    sf_conf = smooth_conf(np.concatenate([sc.sf_seg_conf, sc.sf_seg_conf[dup]]), 0.3)
    new_conf = smooth_conf(sc.tgt_seg_conf, 0.3)
    near = np.argmin(((sc.f64("ed_points")[:, None, :] - P0[None, :, :]) ** 2).sum(-1), axis=1)
    ed_conf = smooth_conf(sc.sf_seg_conf[near], 0.1)
    base.update(num_classes=C, sf_seg=np.argmax(sf_conf, 1).astype(np.int64), sf_seg_conf=sf_conf,
                sf_dist2edge=rs.uniform(0, 20, N), new_seg=np.argmax(new_conf, 1).astype(np.int64),
                new_seg_conf=new_conf, new_dist2edge=rs.uniform(0, 20, T),
                ed_seg=np.argmax(ed_conf, 1).astype(np.int64), ed_seg_conf=ed_conf)
    return base


def run_reference(ref, b, okw):
    t = lambda a: torch.from_numpy(np.array(a, copy=True))     # the reference updates its tensors in place
    opt = SimpleNamespace(height=int(b["H"]), width=int(b["W"]), th_dist=0.1, th_cosine_ang=0.4, th_time_steps=30,
                          disable_merging_new_surfels=False, disable_merging_exist_surfels=False,
                          disable_adding_new_surfels=False, disable_removing_unstable_surfels=False,
                          phase="test", method="super", data="superv2", num_neighbors=int(b["num_neighbors"]), save_sample_freq=10 ** 9,
                          data_dir="")
    for k, v in okw.items():
        if not k.startswith("_"):
            setattr(opt, k, v)
    ed = ref_shim.Data(points=t(b["ed_points"]), radii=t(b["ed_radii"]))
    me = SimpleNamespace(opt=opt, hard_seg=False, evaluate_tracking=False, logger=logging.getLogger("fusion"),
                         points=t(b["sf_points"]), norms=t(b["sf_norms"]), colors=t(b["sf_colors"]), radii=t(b["sf_radii"]),
                         confs=t(b["sf_confs"]), time_stamp=t(b["sf_time_stamp"]), isStable=t(b["sf_isStable"]),
                         knn_indices=t(b["sf_knn_idx"]), knn_w=t(b["sf_knn_w"]), ED_nodes=ed,
                         projdata=torch.zeros(len(b["sf_points"]), 2), summary_writer=ref_shim._SummaryWriter(),
                         render_img=lambda inputs: None, viz=lambda inputs, sfdata: None, time=0)
    inputs = {("color", 0): torch.zeros(1, 3, int(b["H"]), int(b["W"])), "K": torch.from_numpy(b["K"])[None],
              "time": int(b["time"]), "filename": ["%06d" % int(b["time"])]}
    sfdata = ref_shim.Data(points=t(b["new_points"]), norms=t(b["new_norms"]), colors=t(b["new_colors"]),
                           radii=t(b["new_radii"]), confs=t(b["new_confs"]), valid=t(b["new_valid"]),
                           index_map=t(b["new_index_map"]), time=int(b["time"]))
    state = STATE
    if okw.get("_seg"):
        state = STATE + SEG_STATE
        me.hard_seg = bool(okw.get("_hard", False))
        me.power_arg = (1 / 2, 1 / 2)                              # nodes.py:99
        me.seg, me.seg_conf, me.dist2edge = t(b["sf_seg"]), t(b["sf_seg_conf"]), t(b["sf_dist2edge"])
        ed.seg, ed.seg_conf = t(b["ed_seg"]), t(b["ed_seg_conf"])
        sfdata.seg, sfdata.seg_conf, sfdata.dist2edge = t(b["new_seg"]), t(b["new_seg_conf"]), t(b["new_dist2edge"])
    out = {}
    if okw.get("_track"):
        me.track_pts = {}
        me.track_id = t(b["track_id"])
        me.gt, me.gt_strkeys, me.track_rsts = {}, [], {}
        me.update_track_pts = lambda *a, **k: ref.nodes.Surfels.update_track_pts(me, *a, **k)
        me.init_track_pts = lambda *a, **k: ref.nodes.Surfels.init_track_pts(me, *a, **k)
    ref.nodes.Surfels.fuseInputData(me, inputs, sfdata)
    if okw.get("_track"):
        out["fuse_track_id"] = me.track_id.cpu().numpy().copy()
        me.evaluate_tracking = True
    for k in state:
        out["fuse_" + k] = getattr(me, k).detach().cpu().numpy().copy()
    ref.nodes.Surfels.prepareStableIndexNSwapAllModel(me, inputs, sfdata)
    if okw.get("_track"):
        out["swap_track_id"] = me.track_id.cpu().numpy().copy()
    if okw.get("_store_swap", True):
        for k in state:
            out["swap_" + k] = getattr(me, k).detach().cpu().numpy().copy()
    else:
        out["swap_count"] = np.array(len(me.points))
    return out


KNN_MODES = {"plain": dict(), "sem": dict(method="semantic-super"), "hard": dict(method="semantic-super", _hard=True)}


def run_knn(ref, b, okw):
    """``Surfels.update_ed`` + ``update_sfed_knn`` (super/nodes.py:154-191) as run once at frame 0, with the
    Semantic-SuPer branches (class-restricted neighbours, Jensen-Shannon weights)."""
    t = lambda a: torch.from_numpy(np.array(a, copy=True))
    opt = SimpleNamespace(method=okw.get("method", "super"), num_neighbors=int(b["num_neighbors"]), num_ED_neighbors=4,
                          num_classes=int(b["num_classes"]))
    ed = ref_shim.Data(points=t(b["ed_points"]), radii=t(b["ed_radii"]), seg=t(b["ed_seg"]), seg_conf=t(b["ed_seg_conf"]))
    n = len(b["sf_points"])
    me = SimpleNamespace(opt=opt, hard_seg=bool(okw.get("_hard", False)), power_arg=(1 / 2, 1 / 2), ED_nodes=ed,
                         points=t(b["sf_points"]), seg=t(b["sf_seg"]), seg_conf=t(b["sf_seg_conf"]),
                         isStable=torch.ones(n, dtype=torch.bool))
    ref.nodes.Surfels.update_ed(me)
    ref.nodes.Surfels.update_sfed_knn(me)
    return dict(sf_idx=me.knn_indices.numpy(), sf_w=me.knn_w.numpy(), sf_stable=me.isStable.numpy(),
                ed_idx=ed.knn_indices.numpy(), ed_w=ed.knn_w.numpy())


def record(ref, fname, num_neighbors, knn_modes, variants):
    base = make_inputs(num_neighbors=num_neighbors)
    g = {"in_" + k: v for k, v in base.items()}
    for tag in knn_modes:
        for k, v in run_knn(ref, base, KNN_MODES[tag]).items():
            g[f"knn_{tag}_{k}"] = v
        print("knn", tag, "unstable", int((~g[f"knn_{tag}_sf_stable"]).sum()))
    for tag in variants:
        out = run_reference(ref, base, VARIANTS[tag])
        for k, v in out.items():
            g[f"{tag}_{k}"] = v
        print(tag, "surfels", len(base["sf_points"]), "->", len(out["fuse_points"]), "stable", int(out["fuse_isStable"].sum()),
              "->", len(out["swap_points"]) if "swap_points" in out else int(out["swap_count"]))
    path = os.path.join(HERE, fname)
    np.savez_compressed(path, **g)
    print(path, os.path.getsize(path) // 1024, "KB")


def main():
    ref = ref_shim.install()
    record(ref, "fu_48x64.npz", 4, list(KNN_MODES), list(VARIANTS))
    # num_neighbors = 6 (round 6): find_knn / the skinning weights / the candidate search are K-generic in the reference
    # (super/nodes.py:170-191,466-509); the option variants that exercise them
    record(ref, "fu_48x64_k6.npz", 6, ["plain", "sem", "hard"], ["default", "tight", "track", "sem", "hard"])


if __name__ == "__main__":
    main()
