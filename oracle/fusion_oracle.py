"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

NumPy restatement of the reference's surfel fusion step, SURVEY.md 8(f) row f1:
``Surfels.fuseInputData`` (``super/nodes.py:268-541``) and
``Surfels.prepareStableIndexNSwapAllModel`` (``super/nodes.py:543-585``), for ``opt.method ==
"super"`` and ``"semantic-super"`` (segmentation fields carried and fused, Jensen-Shannon skinning
weights, ``hard_seg`` class-restricted neighbours), with or without tracked evaluation points:
project the surfels, build up to 16 confidence-ordered
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
                        phase="test", num_neighbors=4, method="super", data="superv2", num_classes=0,
                        hard_seg=False)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class Model:
    """The surfel arrays ``fuseInputData`` reads and writes (names as in the reference)."""

    def __init__(self, points, norms, colors, radii, confs, time_stamp, isStable, knn_indices, knn_w, ed_points,
                 ed_radii, seg=None, seg_conf=None, dist2edge=None, ed_seg=None, ed_seg_conf=None):
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
        # segmentation fields (Semantic-SuPer): present on the model iff the frames carry them
        self.seg = None if seg is None else np.array(seg, np.int64)
        self.seg_conf = None if seg_conf is None else np.array(seg_conf, np.float64)
        self.dist2edge = None if dist2edge is None else np.array(dist2edge, np.float64)
        self.ed_seg = None if ed_seg is None else np.asarray(ed_seg, np.int64)
        self.ed_seg_conf = None if ed_seg_conf is None else np.asarray(ed_seg_conf, np.float64)


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
    has_seg = m.seg is not None and getattr(d2, "seg", None) is not None
    if (opt.hard_seg or opt.data == "superv1") and has_seg:
        valid &= d1.seg[idx1] == d2.seg[idx2]                    # nodes.py:314-316
    if has_seg:                                                   # read before the rows are overwritten
        sc1, sc2 = d1.seg_conf[idx1][valid], d2.seg_conf[idx2][valid]
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
    if has_seg:                                                   # nodes.py:348-353
        sc = wc.astype(np.float64) * sc1 + wc2.astype(np.float64) * sc2
        sc = sc / sc.sum(1, keepdims=True)
        m.seg_conf[ids] = sc
        m.seg[ids] = np.argmax(sc, axis=1)
    return valid


def kld(P, Q, eps=1e-13):
    """``KLD`` (utils/utils.py:244-250)."""
    return (P * np.log(P / (Q + eps) + eps)).sum(-1)


def jsd(P, Q, eps=1e-13):
    """``JSD`` (utils/utils.py:252-254)."""
    M = 0.5 * (P + Q)
    return 0.5 * (kld(P, M, eps) + kld(Q, M, eps))


def semantic_weights(dist, rad, P, Q):
    """softmax(exp(-JSD)^(1/2) * exp(-d/r)^(1/2)) (nodes.py:183-189,472-477,503-509)."""
    e = np.power(np.exp(-jsd(P, Q)), 0.5) * np.power(np.exp(-dist / rad), 0.5)
    e = np.exp(e - e.max(-1, keepdims=True))
    return e / e.sum(-1, keepdims=True)


def class_knn(points, seg, nodes, node_seg, k, num_classes):
    """``find_knn`` with ``num_classes > 0`` (utils/utils.py:223-242): neighbours among the nodes of
    the point's own class (lowest node index first on ties)."""
    dist = np.full((len(points), k), 1e8)
    idx = -np.ones((len(points), k), np.int64)
    for c in range(num_classes):
        v1 = seg == c
        v2 = np.nonzero(node_seg == c)[0]
        if not v1.any() and len(v2) == 0:
            continue
        assert v1.any() and len(v2) >= k
        d, i = orc.knn(points[v1], nodes[v2], k)
        dist[v1] = d
        idx[v1] = v2[i]
    return dist, idx


def update_ed(ed_points, ed_radii, opt, hard_seg=False, ed_seg=None):
    """``Surfels.update_ed`` (nodes.py:154-168): K_ED nearest other nodes (of the node's own class with
    ``hard_seg``), weights ``softmax(exp(-d / radius_self))``.  Returns (knn_indices, knn_w)."""
    k = opt.num_ED_neighbors + 1
    if hard_seg:
        dist, idx = class_knn(ed_points, ed_seg, ed_points, ed_seg, k, opt.num_classes)
    else:
        dist, idx = orc.knn(ed_points, ed_points, k)
    dist, idx = dist[:, 1:] / np.asarray(ed_radii, np.float64)[:, None], idx[:, 1:]
    e = np.exp(-dist)
    e = np.exp(e - e.max(-1, keepdims=True))
    return idx, e / e.sum(-1, keepdims=True)


def update_sfed_knn(points, is_stable, ed_points, ed_radii, opt, hard_seg=False, seg=None, seg_conf=None, ed_seg=None,
                    ed_seg_conf=None):
    """``Surfels.update_sfed_knn`` (nodes.py:170-191).  Returns (knn_indices, knn_w, isStable)."""
    if hard_seg:
        dist, idx = class_knn(points, seg, ed_points, ed_seg, opt.num_neighbors, opt.num_classes)
    else:
        dist, idx = orc.knn(points, ed_points, opt.num_neighbors)
    rad = np.asarray(ed_radii, np.float64)[idx]
    stable = np.array(is_stable, bool) & (dist <= rad).any(1)
    if opt.method == "semantic-super" and not hard_seg:
        w = semantic_weights(dist, rad, np.asarray(ed_seg_conf)[idx], np.asarray(seg_conf)[:, None, :])
    else:
        w = orc.knn_weights(dist, rad)
    return idx, w, stable


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
    if opt.method == "semantic-super":                            # nodes.py:467-477 (also with hard_seg)
        m.knn_w = semantic_weights(d, m.ed_radii[m.knn_indices], m.ed_seg_conf[m.knn_indices], m.seg_conf[:, None, :])
    else:
        m.knn_w = orc.knn_weights(d, m.ed_radii[m.knn_indices])

    if not opt.disable_adding_new_surfels and add_valid is not None:
        add = add_valid[np.array(new.valid, bool)]
        if add.any():
            pts = new.points[add]
            if opt.hard_seg:                                      # nodes.py:494-497
                dist, idx = class_knn(pts, new.seg[add], m.ed_points, m.ed_seg, opt.num_neighbors, opt.num_classes)
            else:
                dist, idx = orc.knn(pts, m.ed_points, opt.num_neighbors)
            rad = m.ed_radii[idx]
            st = (dist <= rad).any(1)
            if opt.method == "semantic-super" and not opt.hard_seg:
                w = semantic_weights(dist, rad, m.ed_seg_conf[idx], new.seg_conf[add][:, None, :])
            else:
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
            if m.seg is not None:                                 # nodes.py:524-525
                m.seg = np.concatenate([m.seg, new.seg[add][st]])
                m.seg_conf = np.concatenate([m.seg_conf, new.seg_conf[add][st]])
                if m.dist2edge is not None:
                    m.dist2edge = np.concatenate([m.dist2edge, new.dist2edge[add][st]])
    v_, u_, _, _ = project(m.points, K, H, W)
    m.projdata = np.stack([u_, v_], 1).astype(f32)
    return m


def swap_stable(m: Model, opt, time, track_id=None):
    """``prepareStableIndexNSwapAllModel`` (nodes.py:543-590); tracked surfels are kept and renumbered."""
    if not opt.disable_removing_unstable_surfels:
        keep = m.isStable & ((f32(time) - m.time_stamp).astype(f32) < opt.th_time_steps)
        if track_id is not None:
            keep[track_id[track_id >= 0]] = True
        for k in ("points", "norms", "colors", "confs", "radii", "time_stamp", "knn_indices", "knn_w", "projdata",
                  "seg", "seg_conf", "dist2edge"):
            if getattr(m, k) is not None:
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
