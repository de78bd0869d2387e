"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

NumPy restatement of the reference's surfel fusion step, SURVEY.md 8(f) row f1:
``Surfels.fuseInputData`` (``super/nodes.py:268-541``) and
``Surfels.prepareStableIndexNSwapAllModel`` (``super/nodes.py:543-585``), for ``opt.method ==
"super"`` without tracked evaluation points: project the surfels, build up to 16 confidence-ordered
surfel layers per pixel, merge the new frame's points into them, merge surfels that share a pixel,
refresh the skinning weights, append unmatched points as new surfels, then drop unstable / stale
surfels.  The float32 / float64 mix of the reference's tensors is kept operation by operation
(confidences and colours are float32, geometry float64).  Pinned against the reference itself by
``tests/golden/make_golden_fusion.py`` -> ``tests/golden/fu_*.npz``.  Only ``tests/`` import this.

Unpinned by the reference: the order of surfels with EQUAL confidence on the same pixel
(``torch.sort(descending=True)`` is not stable); here the lower index comes first.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np

from oracle import lm_oracle as orc

f32 = np.float32
MAP_NUM = 16


def default_opt(**kw):
    o = SimpleNamespace(height=0, width=0, th_dist=0.1, th_cosine_ang=0.4, th_time_steps=30,
                        disable_merging_new_surfels=False, disable_merging_exist_surfels=False,
                        disable_adding_new_surfels=False, disable_removing_unstable_surfels=False,
                        phase="test", num_neighbors=4)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class Model:
    """The surfel arrays ``fuseInputData`` reads and writes (names as in the reference)."""

    def __init__(self, points, norms, colors, radii, confs, time_stamp, isStable, knn_indices, knn_w, ed_points,
                 ed_radii):
        self.points = np.array(points, np.float64)
        self.norms = np.array(norms, np.float64)
        self.colors = np.array(colors, f32)
        self.radii = np.array(radii, np.float64)
        self.confs = np.array(confs, f32)
        self.time_stamp = np.array(time_stamp, f32)
        self.isStable = np.array(isStable, bool)
        self.knn_indices = np.array(knn_indices, np.int64)
        self.knn_w = np.array(knn_w, np.float64)
        self.ed_points = np.asarray(ed_points, np.float64)
        self.ed_radii = np.asarray(ed_radii, np.float64)
        self.projdata = np.zeros((len(self.points), 2), f32)


def project(points, K, H, W):
    """``pcd2depth`` (utils/utils.py:161-184): float and rounded pixel coordinates, validity."""
    fx, fy, cx, cy = (np.float64(f32(K[0, 0])), np.float64(f32(K[1, 1])), np.float64(f32(K[0, 2])), np.float64(f32(K[1, 2])))
    Z = points[:, 2] + 1e-8
    u_ = points[:, 0] * fx / Z + cx
    v_ = points[:, 1] * fy / Z + cy
    with np.errstate(invalid="ignore"):
        u, v = np.rint(u_).astype(np.int64), np.rint(v_).astype(np.int64)
    coords = v * W + u
    valid = (v >= 0) & (v < H - 1) & (u >= 0) & (u < W - 1)
    return v_, u_, coords, valid


def _normalize(x):
    n = np.sqrt((x * x).sum(-1, keepdims=True))
    return x / np.maximum(n, 1e-12)


def _merge(m, opt, d1, idx1, d2, idx2, time, add_new):
    """``merge_data`` (nodes.py:296-357): fuse rows idx2 of d2 into the surfels idx1 (of d1 == m)."""
    p, n, c, r, w = d1.points[idx1], d1.norms[idx1], d1.colors[idx1], d1.radii[idx1], d1.confs[idx1].astype(f32)
    p2, n2, c2, r2, w2 = d2.points[idx2], d2.norms[idx2], d2.colors[idx2], d2.radii[idx2], d2.confs[idx2].astype(f32)
    if len(p) == 0:
        return np.zeros(0, bool)
    valid = (np.sqrt(((p - p2) ** 2).sum(-1)) < opt.th_dist) & ((n * n2).sum(-1) > opt.th_cosine_ang)
    ids = idx1[valid]
    w, w2 = w[valid], w2[valid]
    wu = (w + w2).astype(f32)
    w = (w / wu).astype(f32)
    w2 = (w2 / wu).astype(f32)
    w64, w264 = w.astype(np.float64)[:, None], w2.astype(np.float64)[:, None]
    m.radii[ids] = w64[:, 0] * r[valid] + w264[:, 0] * r2[valid]
    m.confs[ids] = wu
    m.points[ids] = w64 * p[valid] + w264 * p2[valid]
    m.norms[ids] = _normalize(w64 * n[valid] + w264 * n2[valid])
    wc, wc2 = w[:, None], w2[:, None]
    if add_new:
        wn = (wc2 * f32(3)).astype(f32)
        ws = (wc + wn).astype(f32)
        m.colors[ids] = ((wc / ws).astype(f32) * c[valid]).astype(f32) + ((wn / ws).astype(f32) * c2[valid]).astype(f32)
    else:
        m.colors[ids] = (wc * c[valid]).astype(f32) + (wc2 * c2[valid]).astype(f32)
    if time is not None:
        m.time_stamp[ids] = f32(time)
    return valid


def fuse_input_data(m: Model, opt, K, new, time, track_id=None):
    """``fuseInputData``.  ``new`` has points, norms, colors, radii, confs (T rows), valid (H*W).
    ``track_id`` (tracked evaluation points, modified in place): ids follow the surfel that absorbs
    theirs and become -2 when their surfel is deleted (nodes.py:440-456)."""
    H, W = opt.height, opt.width
    HW = H * W
    valid = np.array(new.valid, bool).copy()
    _, _, coords, val = project(m.points, K, H, W)
    val &= m.isStable
    ids = np.arange(len(m.points))
    # confidence descending (ties: lower index first), then stable by pixel
    conf_order = np.argsort(-m.confs.astype(np.float64), kind="stable")
    coord_order = np.argsort(coords[conf_order], kind="stable")
    order = conf_order[coord_order]
    coords = coords[conf_order][coord_order]
    val = val[order]
    ids = ids[order][val]
    coords = coords[val]
    val = val[val]
    val_maps, index_maps = [], []
    counts = counts_limits = None
    for i in range(MAP_NUM):
        if len(coords) == 0:
            break
        if i == 0:
            tmp, first, cnt = np.unique(coords, return_index=True, return_counts=True)
            counts_limits = np.cumsum(cnt)
            counts = np.concatenate([[0], counts_limits[:-1]])
            sel = counts
            tcoords = tmp
        else:
            counts = counts + 1
            sel = counts[counts < counts_limits]
            tcoords = coords[sel]
        vm = np.zeros(HW, bool)
        vm[tcoords] = True
        im = np.zeros(HW, np.int64)
        im[tcoords] = ids[sel]
        val_maps.append(vm)
        index_maps.append(im)
        val[sel] = False
    left = ids[val]
    del_indices = [left] if len(left) > 0 else []
    t_merge = time if opt.phase == "test" else None

    add_valid = None
    if not opt.disable_merging_new_surfels and val_maps:
        add_valid = valid & ~val_maps[0]
        valid[add_valid] = False
        new_valid_mask = np.array(new.valid, bool)
        for vm, im in zip(val_maps, index_maps):
            if not valid.any():
                break
            v_ = valid & vm
            idx1 = im[v_]
            idx2 = np.nonzero(v_[new_valid_mask])[0]
            mv = _merge(m, opt, m, idx1, new, idx2, t_merge, True)
            valid[v_] = ~mv
        add_valid |= valid

    if not opt.disable_merging_exist_surfels and val_maps:
        n_maps = len(val_maps)
        for i in range(n_maps):
            vm = val_maps[i]                       # the reference ANDs into this array in place
            for j in range(i + 1, n_maps):
                vm &= val_maps[j]
                if not vm.any():
                    continue
                idx1, idx2 = index_maps[i][vm], index_maps[j][vm]
                mv = _merge(m, opt, m, idx1, m, idx2, t_merge, False)
                upd = np.ones(HW, bool)
                upd[vm] = ~mv
                val_maps[j] &= upd
                del_indices.append(idx2[mv])
                if track_id is not None:
                    a, b = idx1[mv], idx2[mv]
                    for k in range(len(track_id)):
                        hit = np.nonzero(b == track_id[k])[0]
                        if len(hit):
                            track_id[k] = a[hit[0]]
        if del_indices:
            dels = np.unique(np.concatenate(del_indices))
            if track_id is not None:
                for k in range(len(track_id)):
                    if track_id[k] in dels:
                        track_id[k] = -2
            m.isStable[dels] = False

    # skinning weights of every surfel at its (possibly fused) position
    d = np.sqrt(((m.points[:, None, :] - m.ed_points[m.knn_indices]) ** 2).sum(-1))
    m.knn_w = orc.knn_weights(d, m.ed_radii[m.knn_indices])

    if not opt.disable_adding_new_surfels and add_valid is not None:
        add = add_valid[np.array(new.valid, bool)]
        if add.any():
            pts = new.points[add]
            dist, idx = orc.knn(pts, m.ed_points, opt.num_neighbors)
            rad = m.ed_radii[idx]
            st = (dist <= rad).any(1)
            w = orc.knn_weights(dist, rad)
            k = int(st.sum())
            m.isStable = np.concatenate([m.isStable, np.ones(k, bool)])
            m.knn_w = np.concatenate([m.knn_w, w[st]])
            m.knn_indices = np.concatenate([m.knn_indices, idx[st]])
            m.points = np.concatenate([m.points, pts[st]])
            m.norms = np.concatenate([m.norms, new.norms[add][st]])
            m.colors = np.concatenate([m.colors, new.colors[add][st].astype(f32)])
            m.radii = np.concatenate([m.radii, new.radii[add][st]])
            m.confs = np.concatenate([m.confs, new.confs[add][st].astype(f32)])
            m.time_stamp = np.concatenate([m.time_stamp, np.full(k, time, f32)])
    v_, u_, _, _ = project(m.points, K, H, W)
    m.projdata = np.stack([u_, v_], 1).astype(f32)
    return m


def swap_stable(m: Model, opt, time, track_id=None):
    """``prepareStableIndexNSwapAllModel`` (nodes.py:543-590); tracked surfels are kept and renumbered."""
    if not opt.disable_removing_unstable_surfels:
        keep = m.isStable & ((f32(time) - m.time_stamp).astype(f32) < opt.th_time_steps)
        if track_id is not None:
            keep[track_id[track_id >= 0]] = True
        for k in ("points", "norms", "colors", "confs", "radii", "time_stamp", "knn_indices", "knn_w", "projdata"):
            setattr(m, k, getattr(m, k)[keep])
        if track_id is not None:
            id_map = -np.ones(len(keep), np.int64)
            id_map[keep] = np.arange(int(keep.sum()))
            live = track_id >= 0
            track_id[live] = id_map[track_id[live]]
        m.isStable = keep[keep]
    if track_id is not None:
        for k in range(len(track_id)):
            if track_id[k] >= 0 and not m.isStable[track_id[k]]:
                track_id[k] = -2
    return m
