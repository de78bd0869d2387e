"""Golden vectors for SURVEY.md 8(f) row f4 (tracking evaluation, host side): run the REFERENCE's ``utils.utils.get_gt``
(``utils/utils.py:360-392``), ``super.nodes.evaluate`` (``super/nodes.py:17-34``) and ``Surfels.init_track_pts`` /
``update_track_pts`` (``super/nodes.py:225-265``), unmodified, through ``ref_shim`` -- with a NON-empty ground truth,
which the ``track`` variant of ``fu_48x64.npz`` does not have (its ``update_track_pts`` returns early).

    python tests/golden/make_golden_track.py        ->  tests/golden/track_48x64.npz

Scene: the surfel model and new frame of ``make_golden_fusion.make_inputs`` (48 x 64).  The synthetic ground truth labels
20 pixels per key frame ``[x, y, visible]``: valid target pixels (some invisible, one on index_map row 0 -- the
reference's ``gt_id > 0`` test drops it --, two on an invalid pixel), and the tracked ids start as a mix of assigned
(>= 0), deleted (-2) and unassigned (-1) points so that every branch of the loop runs.  Stored: the wire-format arrays
``get_gt`` returns, the error vectors of ``evaluate`` (plain / ignored ids / normalised), and ``track_id`` /
``track_rsts`` after ``init_track_pts`` and after three ``update_track_pts`` calls (unknown key frame, a new key frame,
a key frame seen before with moved projections).
"""
from __future__ import annotations

import logging
import os
import sys
import tempfile
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "python-super_amd"))
sys.path.insert(0, HERE)

import ref_shim  # noqa: E402
from make_golden_fusion import make_inputs  # noqa: E402

torch.set_num_threads(1)

KEYS = ["000010", "000020", "000030"]


def make_case(seed=23):
    b = make_inputs()
    rng = np.random.default_rng(seed)
    H, W = int(b["H"]), int(b["W"])
    imap = b["new_index_map"]
    ys, xs = np.nonzero(imap > 0)
    pick = rng.choice(len(ys), 17, replace=False)
    y0, x0 = np.nonzero(imap == 0)                         # the pixel of target row 0: dropped by ``gt_id > 0``
    yi, xi = np.nonzero(imap < 0)
    pts = np.stack([np.concatenate([xs[pick], x0[:1], xi[:2]]), np.concatenate([ys[pick], y0[:1], yi[:2]])], 1)
    assert pts.shape == (20, 2)
    gt = {}
    for n, k in enumerate(KEYS):
        vis = (rng.uniform(size=20) > 0.2).astype(np.float64)
        vis[17] = 1.0                                      # the row-0 pixel IS visible: only ``gt_id > 0`` drops it
        jit = rng.integers(-1, 2, (20, 2)) if n else 0
        xy = np.clip(pts + jit, [0, 0], [W - 1, H - 1]).astype(np.float64)
        gt[k] = np.concatenate([xy, vis[:, None]], 1)
    blob = {"gt": gt, "super_cpp": {k: v + 0.5 for k, v in gt.items()}, "SURF": {k: v - 0.25 for k, v in gt.items()}}
    N = len(b["sf_points"])
    stable = np.nonzero(b["sf_isStable"])[0]
    track_id = np.full(20, -1, dtype=np.int64)
    track_id[[2, 5, 11]] = rng.choice(stable, 3, replace=False)     # already attached
    track_id[[7, 14]] = -2                                          # deleted with their surfel
    projdata = rng.uniform(0, [W, H], (N, 2)).astype(np.float32)
    return b, blob, gt, track_id, projdata


def main():
    ref = ref_shim.install()
    b, blob, gt, track_id, projdata = make_case()
    g = {"track_id0": track_id, "projdata": projdata}
    for k in KEYS:
        g["gtfile_" + k] = gt[k]
    # ---- get_gt: the pickled-dict wire format ----
    with tempfile.TemporaryDirectory() as d:
        np.save(os.path.join(d, "tracked_pts.npy"), blob, allow_pickle=True)
        args = SimpleNamespace(data_dir=d, tracking_gt_file="tracked_pts.npy")
        every, gt_ref, ik, sk, arr = ref.uutils.get_gt(args)
    g["gt_intkeys"], g["gt_strkeys"], g["gt_array"] = np.array(ik), np.array([int(s) for s in sk]), arr
    g["gt_methods"] = np.array([len(every["gt"]), len(every["super_cpp"]), len(every["SURF"])])
    # ---- evaluate ----
    est = gt["000020"].copy()
    est[:, :2] += np.random.default_rng(3).normal(0, 4.0, (20, 2))
    g["eval_est"] = est
    g["eval_plain"] = ref.nodes.evaluate(gt["000020"].copy(), est.copy())
    g["eval_ignored"] = ref.nodes.evaluate(gt["000020"].copy(), est.copy(), igonored_ids=[1, 4, 20])
    g["eval_norm"] = ref.nodes.evaluate(gt["000020"].copy(), est.copy(), normalize=True)
    # ---- init_track_pts / update_track_pts on the fusion scene ----
    t = lambda a: torch.from_numpy(np.array(a, copy=True))
    me = SimpleNamespace(points=t(b["sf_points"]), isStable=t(b["sf_isStable"]), projdata=t(projdata), track_id=t(track_id),
                         track_num=20, gt=gt_ref, gt_strkeys=sk, track_rsts={}, logger=logging.getLogger("track"))
    me.init_track_pts = lambda *a, **k: ref.nodes.Surfels.init_track_pts(me, *a, **k)
    sfdata = ref_shim.Data(points=t(b["new_points"]), index_map=t(b["new_index_map"]))
    ref.nodes.Surfels.init_track_pts(me, sfdata, "000010", th=0.2)
    g["init_track_id"] = me.track_id.numpy().copy()
    g["init_rsts"] = me.track_rsts["000010"].numpy().copy()
    ref.nodes.Surfels.update_track_pts(me, sfdata, "000015")               # not a key frame: nothing happens
    assert set(me.track_rsts) == {"000010"}
    ref.nodes.Surfels.update_track_pts(me, sfdata, "000020", th=0.05)      # a new key frame -> init_track_pts(th)
    g["upd20_track_id"] = me.track_id.numpy().copy()
    g["upd20_rsts"] = me.track_rsts["000020"].numpy().copy()
    moved = projdata[::-1].copy()                                          # the surfels' projections moved
    g["projdata2"] = moved
    me.projdata = t(moved)
    ref.nodes.Surfels.update_track_pts(me, sfdata, "000010")               # seen before: the update loop
    g["upd10_track_id"] = me.track_id.numpy().copy()
    g["upd10_rsts"] = me.track_rsts["000010"].numpy().copy()
    for k in ("sf_points", "sf_isStable", "new_points", "new_index_map", "H", "W"):
        g["in_" + k] = b[k]
    print("attached at init:", int((g["init_track_id"] >= 0).sum()), "of 20; after key frame 20:", int((g["upd20_track_id"] >= 0).sum()))
    path = os.path.join(HERE, "track_48x64.npz")
    np.savez_compressed(path, **g)
    print(path, os.path.getsize(path) // 1024, "KB")


if __name__ == "__main__":
    main()
