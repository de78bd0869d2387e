"""Seeded synthetic scenes in the schema of the reference's hot-path boundary.

No dataset ships with the reference, so parity tests and the
benchmark run on an analytic scene:

* camera: SuPer-V1 intrinsics ``K = [[883,0,445.06],[0,883,190.24]]``
  (reference ``utils/data_loader.py:201-206``), image ``H x W``;
* source surface ``Z0(u,v) = 1 + 0.08 sin(6u/W + phi) cos(5v/H) + 0.05 u/W``;
  the target frame is the same surface with ``phi + dphi``;
* target ("new_data", reference ``utils/data_loader.py:453-489``): one point +
  normal per valid pixel, ``index_map (H,W)`` (-1 invalid) and ``valid (H*W,)``;
* surfels ("sf", reference ``super/nodes.py:135-149``): ``N`` jittered samples
  of the source surface, their K nearest ED nodes and ``softmax(exp(-d/r))``
  weights (reference ``super/nodes.py:170-191``);
* ED nodes (reference ``super/graph_encoder.py:185-192``): a regular grid of
  exactly ``J`` samples of the source surface in row-major order, radii = mean
  distance to the K_ED nearest nodes, node-node KNN (``super/nodes.py:154-168``).

Every array that the HIP path stores as f32 is rounded to f32 here and handed
out both as f32 and as the *same values* in f64, so the f64 reference/oracle
and the f32-storage kernels see identical inputs.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

FX, FY, CX, CY = 883.0, 883.0, 445.06, 190.24  # data_loader.py:203-206


def intrinsics() -> np.ndarray:
    """(4,4) float32 pin-hole matrix exactly as ``SuPerDataset.get_K`` builds it."""
    return np.array([[FX, 0, CX, 0], [0, FY, CY, 0], [0, 0, 1, 0], [0, 0, 0, 1]],
                    dtype=np.float32)


def _scaled_intrinsics(H: int, W: int) -> np.ndarray:
    """Intrinsics rescaled from the native 480x640 to H x W (small test scenes)."""
    K = intrinsics().astype(np.float64)
    K[0, :3] *= W / 640.0
    K[1, :3] *= H / 480.0
    return K.astype(np.float32)


def _surface(u, v, H, W, phi):
    return 1.0 + 0.08 * np.sin(6.0 * u / W + phi) * np.cos(5.0 * v / H) + 0.05 * u / W


def _backproject(u, v, z, K):
    fx, fy, cx, cy = (float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2]))
    return np.stack([(u - cx) * z / fx, (v - cy) * z / fy, z], axis=-1)


def _surface_points_normals(u, v, H, W, phi, K, noise=None):
    """Points on the analytic surface at continuous pixel coords + unit normals
    (central differences of the back-projected surface, facing the camera)."""
    h = 0.5
    z = _surface(u, v, H, W, phi)
    if noise is not None:
        z = z + noise
    p = _backproject(u, v, z, K)
    pu = _backproject(u + h, v, _surface(u + h, v, H, W, phi), K) - \
        _backproject(u - h, v, _surface(u - h, v, H, W, phi), K)
    pv = _backproject(u, v + h, _surface(u, v + h, H, W, phi), K) - \
        _backproject(u, v - h, _surface(u, v - h, H, W, phi), K)
    n = np.cross(pu, pv)
    n /= np.linalg.norm(n, axis=-1, keepdims=True)
    n[n[..., 2] > 0] *= -1.0  # face the camera (camera looks down +Z)
    return p, n


def _grid_shape(J: int, aspect: float):
    best = None
    for gh in range(1, J + 1):
        if J % gh:
            continue
        gw = J // gh
        score = abs(math.log((gh / gw) / aspect))
        if best is None or score < best[0]:
            best = (score, gh, gw)
    return best[1], best[2]


def knn_bruteforce(a: np.ndarray, b: np.ndarray, k: int, chunk: int = 16384):
    """K nearest rows of ``b`` for every row of ``a``: squared L2, ascending,
    ties -> lowest index (the semantics the build pins for
    ``pytorch3d.ops.knn_points``)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    idx = np.empty((len(a), k), np.int64)
    d2o = np.empty((len(a), k), np.float64)
    for s in range(0, len(a), chunk):
        aa = a[s:s + chunk]
        d2 = ((aa[:, None, :] - b[None, :, :]) ** 2).sum(-1)
        order = np.argsort(d2, axis=1, kind="stable")[:, :k]
        idx[s:s + chunk] = order
        d2o[s:s + chunk] = np.take_along_axis(d2, order, axis=1)
    return d2o, idx


def _knn(a, b, k):
    """KNN for scene generation: exact brute force when small, cKDTree otherwise."""
    if len(a) * len(b) <= 4_000_000:
        d2, idx = knn_bruteforce(a, b, k)
        return np.sqrt(d2), idx
    from scipy.spatial import cKDTree
    d, idx = cKDTree(b).query(a, k=k)
    return d, idx.astype(np.int64)


def softmax_exp_weights(dist: np.ndarray, radii: np.ndarray) -> np.ndarray:
    """``softmax(exp(-dist/radius))`` -- a softmax *of an exponential*, literally
    as the reference computes it (``super/nodes.py:166,191``)."""
    e = np.exp(-dist / radii)
    s = np.exp(e - e.max(axis=-1, keepdims=True))
    return s / s.sum(axis=-1, keepdims=True)


def _f32(x):
    return np.ascontiguousarray(np.asarray(x, np.float64).astype(np.float32))


@dataclass
class Scene:
    """One frame pair at the hot-path boundary. ``*_f32``-stored arrays are
    float32; use :meth:`f64` for the identical values widened to float64."""
    H: int
    W: int
    K: np.ndarray                 # (4,4) f32
    sf_points: np.ndarray         # (N,3) f32
    sf_norms: np.ndarray          # (N,3) f32
    sf_knn_idx: np.ndarray        # (N,Kn) i64
    sf_knn_w: np.ndarray          # (N,Kn) f32
    ed_points: np.ndarray         # (J,3) f32
    ed_norms: np.ndarray          # (J,3) f32
    ed_radii: np.ndarray          # (J,) f32
    ed_knn_idx: np.ndarray        # (J,K_ED) i64
    ed_knn_w: np.ndarray          # (J,K_ED) f32
    tgt_points: np.ndarray        # (T,3) f32
    tgt_norms: np.ndarray         # (T,3) f32
    index_map: np.ndarray         # (H,W) i64, -1 invalid
    valid: np.ndarray             # (H*W,) bool
    ed_triangles: np.ndarray = None       # (3,Tr) i64 node-grid triangles (graph_encoder.py:11-67 schema)
    ed_triangle_areas: np.ndarray = None  # (Tr,) f32
    # Semantic-SuPer inputs (make_scene(semantic=True)); data_loader.py:319-331,455-457,494-496
    num_classes: int = 0
    img_seg_conf: np.ndarray = None       # (C,H,W) f32  inputs[("seg_conf",0)][0]
    img_seg: np.ndarray = None            # (H,W) i64    inputs[("seg",0)][0,0]
    tgt_seg_conf: np.ndarray = None       # (T,C) f32    trg.seg_conf (per-pixel softmax of img_seg_conf)
    sf_seg: np.ndarray = None             # (N,) i64     src.seg
    sf_seg_conf: np.ndarray = None        # (N,C) f32    src.seg_conf
    flow: np.ndarray = None               # (1,2,H,W) f32 models.optical_flow(src.rgb, color) (deform_mesh.py:286-320)
    meta: dict = field(default_factory=dict)

    @property
    def N(self):
        return len(self.sf_points)

    @property
    def J(self):
        return len(self.ed_points)

    @property
    def T(self):
        return len(self.tgt_points)

    def f64(self, name):
        return getattr(self, name).astype(np.float64)


def _box_mean(x, k):
    """k x k mean filter over the last two axes, borders averaged over the in-image part
    (``nn.AvgPool2d(k, 1, k//2, count_include_pad=False)``, data_loader.py:322)."""
    r = k // 2
    H, W = x.shape[-2:]
    c = np.zeros(x.shape[:-2] + (H + 1, W + 1))
    c[..., 1:, 1:] = x.cumsum(-2).cumsum(-1)
    y0, y1 = np.clip(np.arange(H) - r, 0, H), np.clip(np.arange(H) + r + 1, 0, H)
    x0, x1 = np.clip(np.arange(W) - r, 0, W), np.clip(np.arange(W) + r + 1, 0, W)
    tot = (c[..., y1[:, None], x1[None, :]] - c[..., y0[:, None], x1[None, :]]
           - c[..., y1[:, None], x0[None, :]] + c[..., y0[:, None], x0[None, :]])
    return tot / ((y1 - y0)[:, None] * (x1 - x0)[None, :])


def _softmax(x, axis):
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    return e / e.sum(axis=axis, keepdims=True)


def _class_logits(u, v, H, W, C, shift):
    """Logits +-4 of C wavy vertical class bands, shifted ``shift`` px to the right."""
    b = np.zeros(np.shape(u), np.int64)
    for c in range(1, C):
        b += (u - shift) >= c * W / C + 0.05 * W * np.sin(2.0 * np.pi * 1.5 * v / H + c)
    return np.where(np.arange(C).reshape((C,) + (1,) * np.ndim(u)) == b[None], 4.0, -4.0)


def make_scene(N=50_000, J=512, H=480, W=640, seed=0, n_neighbors=4, n_ed_neighbors=4,
               phi=0.3, dphi=0.15, src_border=10, tgt_border=4, jitter=0.35,
               depth_noise=1e-4, tgt_holes=0.0, semantic=False, num_classes=3, seg_shift=6.0,
               seg_smooth=5) -> Scene:
    """Build the seeded synthetic frame pair described in the module docstring."""
    rng = np.random.default_rng(seed)
    K = intrinsics() if (H, W) == (480, 640) else _scaled_intrinsics(H, W)
    phi = phi + 0.01 * seed

    # ---- target frame: one point per valid pixel --------------------------------
    vv, uu = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64),
                         indexing="ij")
    tnoise = rng.normal(0.0, depth_noise, size=(H, W)) if depth_noise > 0 else None
    tp, tn = _surface_points_normals(uu, vv, H, W, phi + dphi, K, tnoise)
    valid_map = np.zeros((H, W), bool)
    valid_map[tgt_border:H - tgt_border, tgt_border:W - tgt_border] = True
    if tgt_holes > 0:  # random invalid target pixels -> unmapped bilinear taps
        valid_map &= rng.uniform(size=(H, W)) >= tgt_holes
    index_map = -np.ones((H, W), np.int64)
    index_map[valid_map] = np.arange(int(valid_map.sum()))
    tgt_points = _f32(tp[valid_map])
    tgt_norms = _f32(tn[valid_map])

    # ---- source surfels: N jittered samples of the interior ----------------------
    ih, iw = H - 2 * src_border, W - 2 * src_border
    if N > ih * iw:
        raise ValueError(f"N={N} exceeds the {ih}x{iw} interior; use a larger image")
    pix = np.sort(rng.choice(ih * iw, size=N, replace=False))
    sv = (pix // iw + src_border).astype(np.float64) + rng.uniform(-jitter, jitter, N)
    su = (pix % iw + src_border).astype(np.float64) + rng.uniform(-jitter, jitter, N)
    snoise = rng.normal(0.0, depth_noise, size=N) if depth_noise > 0 else None
    sp, sn = _surface_points_normals(su, sv, H, W, phi, K, snoise)
    sf_points, sf_norms = _f32(sp), _f32(sn)

    # ---- ED nodes: regular grid, row-major ----------------------------------------
    gh, gw = _grid_shape(J, ih / iw)
    gv = src_border + (np.arange(gh) + 0.5) * ih / gh
    gu = src_border + (np.arange(gw) + 0.5) * iw / gw
    gvv, guu = np.meshgrid(gv, gu, indexing="ij")
    ep, en = _surface_points_normals(guu.ravel(), gvv.ravel(), H, W, phi, K)
    ed_points, ed_norms = _f32(ep), _f32(en)

    e64 = ed_points.astype(np.float64)
    dd, ii = _knn(e64, e64, n_ed_neighbors + 1)
    ed_knn_idx = ii[:, 1:]
    ed_radii = _f32(dd[:, 1:].mean(axis=1))
    ed_knn_w = _f32(softmax_exp_weights(dd[:, 1:], ed_radii.astype(np.float64)[:, None]))

    # ---- grid triangles (two per grid cell) and their rest areas (face term) ----------
    gi = np.arange(gh * gw).reshape(gh, gw)
    t0 = np.stack([gi[:-1, :-1], gi[:-1, 1:], gi[1:, :-1]], axis=0).reshape(3, -1)
    t1 = np.stack([gi[:-1, 1:], gi[1:, 1:], gi[1:, :-1]], axis=0).reshape(3, -1)
    ed_triangles = np.concatenate([t0, t1], axis=1).astype(np.int64)
    cr = np.cross(e64[ed_triangles[1]] - e64[ed_triangles[0]], e64[ed_triangles[2]] - e64[ed_triangles[0]])
    ed_triangle_areas = _f32(0.5 * np.sqrt((cr ** 2).sum(1) + 1e-13))

    # ---- surfel -> node KNN + weights ---------------------------------------------
    ds, sf_knn_idx = _knn(sf_points.astype(np.float64), e64, n_neighbors)
    sf_knn_w = _f32(softmax_exp_weights(ds, ed_radii.astype(np.float64)[sf_knn_idx]))

    sem = {}
    if semantic:
        # class bands: the target image is the source segmentation moved seg_shift px
        C = num_classes
        img_conf = _f32(_box_mean(_class_logits(uu, vv, H, W, C, seg_shift), seg_smooth))
        src_conf = _box_mean(_class_logits(uu, vv, H, W, C, 0.0), seg_smooth)
        pv = np.clip(np.rint(sv).astype(np.int64), 0, H - 1)
        pu = np.clip(np.rint(su).astype(np.int64), 0, W - 1)
        sf_conf = _f32(_softmax(src_conf[:, pv, pu].T, 1))
        sem = dict(num_classes=C, img_seg_conf=img_conf,
                   img_seg=np.argmax(img_conf, axis=0).astype(np.int64),
                   tgt_seg_conf=_f32(_softmax(img_conf.astype(np.float64), 0).transpose(1, 2, 0)[valid_map]),
                   sf_seg=np.argmax(sf_conf, axis=1).astype(np.int64), sf_seg_conf=sf_conf)

    return Scene(**sem, H=H, W=W, K=K, sf_points=sf_points, sf_norms=sf_norms,
                 sf_knn_idx=np.ascontiguousarray(sf_knn_idx, dtype=np.int64), sf_knn_w=sf_knn_w,
                 ed_points=ed_points, ed_norms=ed_norms, ed_radii=ed_radii,
                 ed_knn_idx=np.ascontiguousarray(ed_knn_idx, dtype=np.int64), ed_knn_w=ed_knn_w,
                 tgt_points=tgt_points, tgt_norms=tgt_norms, index_map=index_map,
                 valid=valid_map.reshape(-1).copy(),
                 ed_triangles=ed_triangles, ed_triangle_areas=ed_triangle_areas,
                 meta=dict(N=N, J=J, H=H, W=W, seed=seed, grid=(gh, gw), phi=phi, dphi=dphi))


# Named workloads of BASELINE.json `configs` (C1, C2, C4 geometry).
WORKLOADS = {
    "C1": dict(N=50_000, J=512, H=480, W=640),
    "C2": dict(N=200_000, J=2_000, H=480, W=640),
    "C4": dict(N=500_000, J=4_000, H=720, W=960),
}


def smooth_flow(H, W, seed, amp=(1.2, 0.9)):
    """A smooth synthetic optical-flow field (1,2,H,W) float32, channel 0 = x (u) displacement, channel 1 = y,
    of up to ``amp`` pixels: stands in for the output of the reference's flow network (``models.optical_flow``,
    ``deform_mesh.py:286-320``), which is outside the hot path."""
    rng = np.random.default_rng(seed)
    vv, uu = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    a = rng.uniform(0.5, 1.5, 4)
    fx = amp[0] * np.sin(2 * np.pi * a[0] * uu / W + 0.3) * np.cos(2 * np.pi * a[1] * vv / H)
    fy = amp[1] * np.cos(2 * np.pi * a[2] * uu / W) * np.sin(2 * np.pi * a[3] * vv / H + 0.7)
    return np.stack([fx, fy])[None].astype(np.float32)


def graphfit_options(**kw):
    """An options object for ``GraphFit`` with the reference's option NAMES (``options.py:37-50,213-250,331-343``) and the
    values the benchmarks and the synthetic driver runs use: point-plane + ARAP (weight 10) + Rot, 10 optimiser
    iterations at learning rate 5e-5, no face / segmentation / morphing / correspondence term.  Product-side helper
    (``bench.py`` and the tools build their GraphFit runs from it); the test oracle carries its own defaults."""
    from types import SimpleNamespace
    o = SimpleNamespace(sf_point_plane=True, sf_point_plane_weight=1.0, mesh_arap=True, mesh_arap_weight=10.0,
                        mesh_rot=True, mesh_rot_weight=1.0, mesh_face=False, mesh_face_weight=1.0,
                        num_optimize_iterations=10, optimizer="SGD", learning_rate=5e-5,
                        sf_soft_seg_point_plane=False, sf_hard_seg_point_plane=False, sf_bn_morph=False,
                        sf_bn_morph_weight=1.0, depth_model="monodepth2", sf_corr=False, sf_corr_weight=0.001,
                        sf_corr_loss_type="point-point", deform_udpate_method="super_edg")
    for k, v in kw.items():
        setattr(o, k, v)
    return o
