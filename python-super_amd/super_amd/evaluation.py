"""Tracking evaluation, SURVEY.md 8(f) row f4: the ground-truth wire format (``get_gt``
``utils/utils.py:360-392``), the reprojection error (``evaluate`` ``super/nodes.py:17-34``) and the
tracked-point bookkeeping of ``Surfels`` (``init_track_pts`` / ``update_track_pts``
``super/nodes.py:225-265``).  Host-side: 20 labelled points per key frame.

Ground-truth file (``opt.tracking_gt_file``): a pickled dict saved with ``np.save`` --
``{"gt": {"000010": (20,3) array [x, y, visible], ...}, "super_cpp": {...}, "SURF": {...}}``.
"""
from __future__ import annotations

import os

import numpy as np
import torch


def get_gt(args):
    """Returns ``(all_methods, gt, gt_intkeys, gt_strkeys, gt_array)`` like the reference."""
    data_dir = os.path.expanduser(args.data_dir)
    if not os.path.exists(data_dir):
        raise ValueError(f"Path {data_dir} does not exist. This is likely an error with args.data_dir configuration.")
    path = os.path.join(data_dir, args.tracking_gt_file)
    if not os.path.exists(path):
        raise ValueError("Ground truth file does not exist!")
    everything = np.array(np.load(path, allow_pickle=True)).tolist()
    gt = everything["gt"]
    int_keys = sorted(int(k) for k in gt.keys())
    str_keys = sorted(f"{int(k):06d}" for k in gt.keys())
    return everything, gt, int_keys, str_keys, np.stack([gt[k] for k in str_keys], axis=0)


def evaluate(gt, est, igonored_ids=(), normalize=False):
    """Per-point pixel distance between labelled and tracked points; -1 where the label is not visible
    (or the 1-based id is ignored); divided by the image height 480 when ``normalize``."""
    gt, est = np.asarray(gt, dtype=np.float64), np.asarray(est, dtype=np.float64)
    seen = gt[:, 2] == 1
    if len(igonored_ids) > 0:
        seen[np.asarray(igonored_ids) - 1] = False
    d = np.sqrt(((gt[:, :2] - est[:, :2]) ** 2).sum(1))
    d[~seen] = -1
    return d / 480 if normalize else d


def init_track_pts(sf, sfdata, filename, th=0.2):
    """Attach still-unassigned labelled points (track_id == -1) to the nearest stable surfel of the
    pixel they label (within ``th``), then record every point's current projection."""
    if filename not in sf.gt:
        return
    dev = sf.points.device
    sf.track_rsts[filename] = torch.zeros((sf.track_num, 3), device=dev)
    labels = torch.as_tensor(sf.gt[filename], dtype=torch.int)
    for k in range(len(sf.track_id)):
        tid = sf.track_id[k]
        x, y, seen = (int(v) for v in labels[k])
        row = int(sfdata.index_map[y, x])
        if int(tid) < 0 and row > 0 and seen == 1:
            d = torch.linalg.norm(sf.points - sfdata.points[row], dim=-1)
            taken = sf.track_id[(sf.track_id >= 0) | (sf.track_id == -2)]
            if len(taken) > 0:
                d[taken.to(torch.long)] = 1e13
                d[~sf.isStable] = 1e13
            if float(d.min()) < th:
                sf.track_id[k] = int(torch.argmin(d))
        # ``tid`` is a 0-d VIEW of track_id[k] (the reference iterates the tensor: ``for k, tid in enumerate(self.track_id)``),
        # so it shows the id assigned just above; a point that stays unassigned indexes projdata[-1] / [-2] like the reference
        # (pinned by tests/golden/track_48x64.npz, recorded from the reference with a non-empty gt)
        sf.track_rsts[filename][k, 0:2] = sf.projdata[int(tid)]
        sf.track_rsts[filename][k, 2] = 1


def update_track_pts(sf, sfdata, filename, th=1e-2):
    if filename not in set(sf.gt_strkeys):
        return
    if filename not in sf.track_rsts:
        init_track_pts(sf, sfdata, filename, th)
        return
    for k in range(len(sf.track_id)):
        tid = int(sf.track_id[k])
        if tid < 0:
            continue
        sf.track_rsts[filename][k, 0:2] = sf.projdata[tid]
        sf.track_rsts[filename][k, 2] = 1
