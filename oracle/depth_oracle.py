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

PARITY UNPINNED for one function: ``skimage_ssim_full`` restates the published algorithm of
``skimage.metrics.structural_similarity`` (scikit-image, a dependency absent from the reference tree and
from this image, and not version-pinned by the reference: ``resources/environment.yaml:28``); the stereo
warp before it and the confidence blend after it ARE pinned by the reference (``v2ssim`` golden variant,
recorded with this restatement standing in for the missing package).
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


def skimage_ssim_full(im1, im2, win_size=7, data_range=2.0, K1=0.01, K2=0.03):
    """Full SSIM image of ``skimage.metrics.structural_similarity(im1, im2, channel_axis=0, full=True)``
    as the reference calls it (utils/data_loader.py:368-371).  scikit-image is a dependency that is
    absent from /root/reference and from this image and is NOT version-pinned by the reference
    (``resources/environment.yaml:28``); this restates the published algorithm of the 0.19 line
    (Wang et al. 2004 with a 7x7 uniform window, sample covariance, ``scipy.ndimage.uniform_filter``
    with reflecting borders, float32 images, and -- no ``data_range`` given -- the float dtype range
    -1..1, i.e. ``data_range = 2``).  PARITY UNPINNED for this function: no skimage output is available
    to check it against.  im1, im2: (C,H,W) float32."""
    from scipy.ndimage import uniform_filter
    C1, C2 = f32((K1 * data_range) ** 2), f32((K2 * data_range) ** 2)
    NP = win_size ** 2
    cov_norm = f32(NP / (NP - 1))
    S = np.empty(im1.shape, f32)
    for ch in range(im1.shape[0]):
        x, y = im1[ch].astype(f32), im2[ch].astype(f32)
        ux, uy = uniform_filter(x, size=win_size), uniform_filter(y, size=win_size)
        uxx, uyy, uxy = uniform_filter(x * x, size=win_size), uniform_filter(y * y, size=win_size), uniform_filter(x * y, size=win_size)
        vx, vy, vxy = cov_norm * (uxx - ux * ux), cov_norm * (uyy - uy * uy), cov_norm * (uxy - ux * uy)
        A1, A2, B1, B2 = 2 * ux * uy + C1, 2 * vxy + C2, ux ** 2 + uy ** 2 + C1, vx + vy + C2
        S[ch] = (A1 * A2) / (B1 * B2)
    return S


def project3d_grid(depth, inv_K, K, stereo_T):
    """``BackprojectDepth`` + ``Project3D`` (depth/monodepth2/layers.py:141-192) in float32: the
    sampling grid in [-1, 1] at which the reference warps the image (data_loader.py:362-363)."""
    H, W = depth.shape
    cam = backproject(depth.astype(f32), inv_K).reshape(-1, 3)              # (HW,3) float32
    P = (np.asarray(K, f32) @ np.asarray(stereo_T, f32))[:3, :].astype(f32)
    q = (cam[:, 0:1] * P[:, 0] + cam[:, 1:2] * P[:, 1] + cam[:, 2:3] * P[:, 2] + P[:, 3]).astype(f32)
    den = (q[:, 2] + f32(1e-7)).astype(f32)
    gx = ((q[:, 0] / den / f32(W - 1) - f32(0.5)) * f32(2)).astype(f32)
    gy = ((q[:, 1] / den / f32(H - 1) - f32(0.5)) * f32(2)).astype(f32)
    return gx.reshape(H, W), gy.reshape(H, W)


def grid_sample_bilinear(img, gx, gy):
    """``F.grid_sample(img, grid)`` defaults: bilinear, zeros padding, align_corners=False.
    img (C,H,W) float32, grid in [-1,1]."""
    C, H, W = img.shape
    ix = (((gx + f32(1)) * f32(W) - f32(1)) / f32(2)).astype(f32)
    iy = (((gy + f32(1)) * f32(H) - f32(1)) / f32(2)).astype(f32)
    with np.errstate(invalid="ignore"):
        x0, y0 = np.floor(ix), np.floor(iy)
    wx1, wy1 = (ix - x0).astype(f32), (iy - y0).astype(f32)
    wx0, wy0 = (f32(1) - wx1).astype(f32), (f32(1) - wy1).astype(f32)
    out = np.zeros((C,) + gx.shape, f32)
    bad = ~(np.isfinite(ix) & np.isfinite(iy))
    for dy, wy in ((0, wy0), (1, wy1)):
        for dx, wx in ((0, wx0), (1, wx1)):
            xx = np.where(bad, -1, x0 + dx).astype(np.int64)
            yy = np.where(bad, -1, y0 + dy).astype(np.int64)
            ok = (xx >= 0) & (xx < W) & (yy >= 0) & (yy < H)
            v = img[:, np.clip(yy, 0, H - 1), np.clip(xx, 0, W - 1)]
            out += np.where(ok, (wx * wy).astype(f32), f32(0))[None] * v
    out[:, bad] = np.nan
    return out


def stereo_confidence(depth, K, inv_K, stereo_T, color):
    """``inputs[("disp_conf",0)]`` (data_loader.py:359-373): SSIM between the image and its warp through
    the stereo transform, mean over the colour channels.  (H,W) float32."""
    gx, gy = project3d_grid(depth, inv_K, K, stereo_T)
    warp = grid_sample_bilinear(np.asarray(color, f32), gx, gy)
    return skimage_ssim_full(warp, np.asarray(color, f32)).mean(0).astype(f32)


def depth_preprocessing(opt, depth, K, inv_K, color, divterm, seg=None, seg_conf=None, valid_mask=None,
                        stereo_T=None):
    """Returns the fields of the reference's ``Data`` object as NumPy arrays (float64 where the
    reference hands out float64).  ``stereo_T`` (4,4) with ``opt.disable_ssim_conf == False``: the
    confidence is blended with the stereo SSIM confidence (data_loader.py:477-479)."""
    H, W = depth.shape
    depth = depth.astype(f32).copy()
    disp_conf = None
    if hasattr(opt, "disable_ssim_conf") and not opt.disable_ssim_conf:
        disp_conf = stereo_confidence(depth, K, inv_K, stereo_T, color)
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
    if disp_conf is not None:
        sig = (f32(1) / (f32(1) + np.exp(-disp_conf).astype(f32))).astype(f32)
        confs = (f32(0.5) * confs + f32(0.5) * sig).astype(f32)
    out = dict(points=pts, norms=nrm, colors=np.transpose(color, (1, 2, 0))[valid], radii=radii, confs=confs[valid],
               valid=valid.reshape(-1), index_map=index_map, valid_map=valid, inval=inval)
    if disp_conf is not None:
        out["disp_conf"] = disp_conf
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
