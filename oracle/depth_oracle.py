"""ORACLE -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

NumPy restatement of the reference's ``depth_preprocessing`` (``utils/data_loader.py:333-523``,
with ``getN`` ``:532-584``, ``BackprojectDepth`` ``depth/monodepth2/layers.py:139-167``,
``torch_dilate`` ``utils/utils.py:152-157`` and ``find_edge_region`` ``utils/utils.py:276-301``):
the step that turns a depth map into the per-frame target ``new_data`` (points, normals, colours,
radii, confidences, ``index_map``, ``valid``, optional semantic fields) which the hot path consumes.
SURVEY.md 8(f) row f2.

Float32 arithmetic is reproduced operation by operation where the reference computes in float32
(back-projection = ``a0*b0`` then two fused multiply-adds, exactly what ``torch.matmul`` does for
the 3x3 intrinsics here; point differences, cross products, normalisation), so points are
bit-exact and normals agree to float32 rounding.  Pinned against the reference itself by
``tests/golden/make_golden_depth.py`` -> ``tests/golden/dp_*.npz``.  Only ``tests/`` import this.
"""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np

f32 = np.float32


def default_opt(**kw):
    o = SimpleNamespace(height=0, width=0, data="superv1", load_valid_mask=False, depth_model="monodepth2",
                        dilate_invalid_kernel=0, normal_model="naive", phase="test", load_depth=True,
                        depth_width_range=(0.02, 0.98), del_seg_classes=(), num_classes=3)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def _fma32(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def backproject(depth, inv_K):
    """``BackprojectDepth.forward`` in float32: inv_K[:3,:3] @ [u, v, 1] (one rounded product, two
    fused multiply-adds, in k order) times the depth.  Returns (H,W,3) float32."""
    H, W = depth.shape
    u = np.broadcast_to(np.arange(W, dtype=f32)[None, :], (H, W))
    v = np.broadcast_to(np.arange(H, dtype=f32)[:, None], (H, W))
    one = np.ones((H, W), f32)
    iK = np.asarray(inv_K, f32)[:3, :3]
    out = []
    for i in range(3):
        acc = (np.full((H, W), iK[i, 0], f32) * u).astype(f32)
        acc = _fma32(np.full((H, W), iK[i, 1], f32), v, acc)
        acc = _fma32(np.full((H, W), iK[i, 2], f32), one, acc)
        out.append((depth.astype(f32) * acc).astype(f32))
    return np.stack(out, -1)


def dilate(mask, k):
    """``torch_dilate``: k x k box 'same' convolution > 0 (PyTorch pads (k-1)//2 before, the rest after)."""
    H, W = mask.shape
    lo = (k - 1) // 2
    hi = k - 1 - lo
    p = np.pad(mask.astype(bool), ((lo, hi), (lo, hi)), constant_values=False)
    out = np.zeros((H, W), bool)
    for dy in range(k):
        for dx in range(k):
            out |= p[dy:dy + H, dx:dx + W]
    return out


def invalid_map(opt, depth, seg=None, valid_mask=None):
    """Step 1 of depth_preprocessing (data_loader.py:375-436): the pixels to drop."""
    H, W = depth.shape
    if opt.data == "superv1":
        inval = ~valid_mask.astype(bool) if (opt.load_valid_mask and valid_mask is not None) else np.zeros((H, W), bool)
        for c in getattr(opt, "del_seg_classes", ()) or ():
            inval |= seg == c
        k = int(opt.dilate_invalid_kernel)
        if opt.depth_model == "raft_stereo":
            if k > 0:
                inval = dilate(inval, k)
            inval[:, :int(0.05 * W)] = True
        elif k > 0:
            inval = ~dilate(~inval, k)
            inval = dilate(inval, 2 * k)
        inval |= depth <= 0
        inval |= depth > 1.5
    elif opt.data == "superv2":
        inval = np.zeros((H, W), bool)
        if opt.load_depth:
            inval |= depth == 0
            # quirk kept: the reference slices dim 2 of a (1,1,H,W) tensor, i.e. ROWS (data_loader.py:411-412)
            inval[0:int(0.1 * W), :] = True
        else:
            inval[0:int(opt.depth_width_range[0] * W), :] = True
            inval[int(opt.depth_width_range[1] * W):, :] = True
        for c in getattr(opt, "del_seg_classes", ()) or ():
            inval |= seg == c
    else:
        raise ValueError(opt.data)
    return inval


def _normalize32(n):
    """``F.normalize(N, dim=-1)`` in float32: N / max(||N||_2, 1e-12)."""
    sq = (n * n).astype(f32)
    nrm = np.sqrt(((sq[..., 0] + sq[..., 1]).astype(f32) + sq[..., 2]).astype(f32)).astype(f32)
    return (n / np.maximum(nrm, f32(1e-12))[..., None]).astype(f32)


def _cross32(a, b):
    return np.stack([(a[..., 1] * b[..., 2]).astype(f32) - (a[..., 2] * b[..., 1]).astype(f32),
                     (a[..., 2] * b[..., 0]).astype(f32) - (a[..., 0] * b[..., 2]).astype(f32),
                     (a[..., 0] * b[..., 1]).astype(f32) - (a[..., 1] * b[..., 0]).astype(f32)], -1).astype(f32)


def normals(points, colors=None):
    """``getN`` (data_loader.py:532-584) in float32 on a NaN-padded vertex map.  colors (3,H,W)
    selects the colour-weighted 8-neighbour variant.  Returns (N (H,W,3) f32, valid (H,W) bool)."""
    H, W, _ = points.shape
    P = np.pad(points.astype(f32), ((1, 1), (1, 1), (0, 0)), constant_values=np.nan)
    sh = lambda dy, dx: P[1 + dy:1 + dy + H, 1 + dx:1 + dx + W]
    with np.errstate(invalid="ignore", divide="ignore"):
        if colors is None:
            hL, hR, hD, hU = sh(0, -1), sh(0, 1), sh(-1, 0), sh(1, 0)     # names as in the reference
            N = _cross32((hR - hL).astype(f32), (hD - hU).astype(f32))
        else:
            C = np.pad(np.transpose(colors.astype(f32), (1, 2, 0)), ((1, 1), (1, 1), (0, 0)), constant_values=np.nan)
            cs = lambda dy, dx: C[1 + dy:1 + dy + H, 1 + dx:1 + dx + W]
            cen_c, cen = cs(0, 0), sh(0, 0)

            def wgt(dy, dx):
                d = np.abs((cs(dy, dx) - cen_c).astype(f32))
                m = (((d[..., 0] + d[..., 1]).astype(f32) + d[..., 2]).astype(f32) / f32(3)).astype(f32)
                return np.exp(-m).astype(f32)[..., None]

            offs = dict(L=(0, -1), LU=(-1, -1), U=(-1, 0), RU=(-1, 1), R=(0, 1), RD=(1, 1), D=(1, 0), DL=(1, -1))
            h = {k: ((sh(*o) - cen).astype(f32) * wgt(*o)).astype(f32) for k, o in offs.items()}
            order = ["L", "LU", "U", "RU", "R", "RD", "D", "DL"]
            N = np.zeros((H, W, 3), f32)
            for a in range(7):
                rest = h[order[a + 1]]
                for b in range(a + 2, 8):
                    rest = (rest + h[order[b]]).astype(f32)
                N = (N + _cross32(h[order[a]], rest)).astype(f32)
        N = _normalize32(N)
    return N, ~np.isnan(N).any(-1)


def edge_points_norm(img_seg, num_classes, kernel=3):
    """Per class the boundary pixels (x / W, y / H) as float64 (data_loader.py:498-510)."""
    seg = np.asarray(img_seg)
    H, W = seg.shape
    out = []
    for c in range(num_classes):
        m = seg == c
        near = dilate(~m, kernel) if kernel % 2 == 1 else None
        e = near & m
        e[:kernel] = e[-kernel:] = False
        e[:, :kernel] = e[:, -kernel:] = False
        ey, ex = np.nonzero(e)
        # integer / int is a float32 division in the reference, widened to float64 afterwards
        out.append(np.stack([(ex.astype(f32) / f32(W)).astype(f32), (ey.astype(f32) / f32(H)).astype(f32)], 1)
                   .astype(np.float64).reshape(-1, 2))
    return out


def depth_preprocessing(opt, depth, K, inv_K, color, divterm, seg=None, seg_conf=None, valid_mask=None):
    """Returns the fields of the reference's ``Data`` object as NumPy arrays (float64 where the
    reference hands out float64)."""
    H, W = depth.shape
    depth = depth.astype(f32).copy()
    pcd = backproject(depth, inv_K)
    inval = invalid_map(opt, depth, seg, valid_mask)
    depth[inval] = np.nan
    pcd[inval] = np.nan
    N, valid = normals(pcd, color if opt.normal_model == "8neighbors" else None)
    valid &= ~np.isnan(pcd).any(-1)
    Z = -depth                                           # quirk kept: Z = -depth (data_loader.py:447)
    index_map = -np.ones((H, W), np.int64)
    index_map[valid] = np.arange(int(valid.sum()))
    pts = pcd[valid].astype(np.float64)
    nrm = N[valid].astype(np.float64)
    # radii: float32 depth / (float64 sqrt(2) * float32 fx * float64 clamp) -> float64
    radii = Z[valid].astype(np.float64) / (np.sqrt(2) * np.float64(f32(K[0, 0])) * np.clip(np.abs(nrm[:, 2]), 0.26, 1.0))
    # confidence: integer grids divided by Python ints give float32 tensors in the reference
    U, V = np.meshgrid(np.arange(W), np.arange(H), indexing="xy")
    su, sv = (U.astype(f32) / f32(W)).astype(f32), (V.astype(f32) / f32(H)).astype(f32)
    a, b = (f32(2) * su - f32(1)).astype(f32), (f32(2) * sv - f32(1)).astype(f32)
    dc2 = ((a * a).astype(f32) + (b * b).astype(f32)).astype(f32)
    confs = np.exp((-dc2 * f32(divterm)).astype(f32)).astype(f32)
    out = dict(points=pts, norms=nrm, colors=np.transpose(color, (1, 2, 0))[valid], radii=radii, confs=confs[valid],
               valid=valid.reshape(-1), index_map=index_map, valid_map=valid, inval=inval)
    if seg is not None:
        sc = np.exp(seg_conf - seg_conf.max(0, keepdims=True))
        sc = (sc / sc.sum(0, keepdims=True)).transpose(1, 2, 0)
        out["seg"] = seg[valid]
        out["seg_conf"] = sc[valid]
        E = edge_points_norm(seg, opt.num_classes)
        # pcd2depth(round_coords=False) on float64 points with the float32 intrinsics
        fx, fy, cx, cy = (np.float64(f32(K[0, 0])), np.float64(f32(K[1, 1])), np.float64(f32(K[0, 2])), np.float64(f32(K[1, 2])))
        Zp = pts[:, 2] + 1e-8
        sx, sy = pts[:, 0] * fx / Zp + cx, pts[:, 1] * fy / Zp + cy
        sf = np.stack([sx / W, sy / H], 1)
        d2e = np.zeros(len(pts))
        for c in range(opt.num_classes):
            sel = out["seg"] == c
            if sel.any() and len(E[c]):
                d2 = ((sf[sel][:, None, :] - E[c][None, :, :]) ** 2).sum(-1)
                d2e[sel] = np.sqrt(d2.min(1))
        out["dist2edge"] = d2e
    return out
