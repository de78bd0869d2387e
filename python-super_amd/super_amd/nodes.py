"""Host-side mirrors of the ``Surfels`` methods on the hot path (``super/nodes.py``):

* :func:`update`          <- ``Surfels.update``            (``super/nodes.py:193-223``)
* :func:`update_sfed_knn` <- ``Surfels.update_sfed_knn``   (``super/nodes.py:170-191``)
* :func:`update_ed`       <- ``Surfels.update_ed``         (``super/nodes.py:154-168``)
* :func:`find_knn`        <- ``utils.utils.find_knn``      (``utils/utils.py:212-242``)

``update_ed`` / ``update_sfed_knn`` follow the Semantic-SuPer branches too: with ``sf.hard_seg`` the
neighbours come from the nodes of the point's own class, with ``opt.method == "semantic-super"`` (and no
``hard_seg``) the surfel weights carry the Jensen-Shannon factor (``super/nodes.py:157-160,172-189``).

Each takes the reference's ``sf`` object (anything with the same attributes), runs the
HIP kernels through the C ABI and writes the results back with the reference's
attribute names and dtypes.  They can be bound onto the reference class:
``Surfels.update = super_amd.nodes.update``.
"""
from __future__ import annotations

import torch

from . import _lib
from .LM import _as, _dev_ptr, _stream_ptr


def find_knn(points1, points2, num_classes=-1, seg1=None, seg2=None, k=20, skip_self=False):
    """K nearest rows of ``points2`` for each row of ``points1``: returns
    ``(dists (N,k) float64 = sqrt(d2), idx (N,k) int64)``, squared-L2 ascending,
    ties -> lowest index.  ``num_classes > 0`` with ``seg1`` / ``seg2``: neighbours among the rows of
    the query's own class (float64 kernel; a class with fewer than k rows raises like the reference's
    assert)."""
    lib = _lib.load()
    dev = points1.device
    if num_classes > 0 or points1.dtype == torch.float64 or points2.dtype == torch.float64:
        # float64 like the reference's tensors (distances, and so the stability test
        # dist <= radius and the weights, are not rounded)
        q = _as(points1, torch.float64, dev)
        n = _as(points2, torch.float64, dev)
        s1 = s2 = None
        if num_classes > 0:
            s1, s2 = _as(seg1, torch.int32, dev), _as(seg2, torch.int32, dev)
        idx = torch.empty((q.shape[0], k), dtype=torch.int32, device=dev)
        dist = torch.empty((q.shape[0], k), dtype=torch.float64, device=dev)
        _lib.check(lib.slm_knn_f64(q.shape[0], n.shape[0], k, int(skip_self), _dev_ptr(q), _dev_ptr(n),
                                   _dev_ptr(s1) if s1 is not None else None, _dev_ptr(s2) if s2 is not None else None,
                                   _dev_ptr(idx), _dev_ptr(dist), _stream_ptr(dev)), "slm_knn_f64")
        return dist, idx.to(torch.int64)
    # float32 state (the compact layout): float32 storage of the distances
    q = _as(points1, torch.float32, dev)
    n = _as(points2, torch.float32, dev)
    idx = torch.empty((q.shape[0], k), dtype=torch.int32, device=dev)
    dist = torch.empty((q.shape[0], k), dtype=torch.float32, device=dev)
    _lib.check(lib.slm_knn(q.shape[0], n.shape[0], k, int(skip_self), _dev_ptr(q), _dev_ptr(n),
                           _dev_ptr(idx), _dev_ptr(dist), _stream_ptr(dev)), "slm_knn")
    return dist, idx.to(torch.int64)


def _weights(idx, dist, radii, radius_mode, stable=None):
    lib = _lib.load()
    dev = idx.device
    idx32 = _as(idx, torch.int32, dev)
    d32 = _as(dist, torch.float32, dev)
    r32 = _as(radii, torch.float32, dev)
    w = torch.empty(idx.shape, dtype=torch.float32, device=dev)
    st8 = None
    if stable is not None:
        st8 = _as(stable, torch.uint8, dev)
    _lib.check(lib.slm_knn_weights(idx.shape[0], idx.shape[1], radius_mode, _dev_ptr(idx32),
                                   _dev_ptr(d32), _dev_ptr(r32), _dev_ptr(w),
                                   _dev_ptr(st8) if st8 is not None else None,
                                   _stream_ptr(dev)), "slm_knn_weights")
    return w, st8


def _weights64(idx, dist, radii, radius_mode, stable=None, q_conf=None, node_conf=None):
    """float64 weights, optionally with the Jensen-Shannon factor of Semantic-SuPer."""
    lib = _lib.load()
    dev = idx.device
    idx32 = _as(idx, torch.int32, dev)
    d = _as(dist, torch.float64, dev)
    r = _as(radii, torch.float64, dev)
    w = torch.empty(idx.shape, dtype=torch.float64, device=dev)
    st8 = _as(stable, torch.uint8, dev).clone() if stable is not None else None
    C_ = 0
    if q_conf is not None:
        q_conf, node_conf = _as(q_conf, torch.float64, dev), _as(node_conf, torch.float64, dev)
        C_ = int(q_conf.shape[1])
    _lib.check(lib.slm_knn_weights_f64(idx.shape[0], idx.shape[1], radius_mode, _dev_ptr(idx32), _dev_ptr(d),
                                       _dev_ptr(r), C_, _dev_ptr(q_conf) if C_ else None,
                                       _dev_ptr(node_conf) if C_ else None, _dev_ptr(w),
                                       _dev_ptr(st8) if st8 is not None else None, _stream_ptr(dev)),
               "slm_knn_weights_f64")
    return w, st8


def update_ed(sf):
    """Node-node KNN + ``softmax(exp(-dist/radius_self))`` weights (K_ED+1 nearest, self
    dropped); with ``sf.hard_seg`` among the nodes of the node's own class."""
    ed = sf.ED_nodes
    k = int(sf.opt.num_ED_neighbors)
    if getattr(sf, "hard_seg", False):
        dist, idx = find_knn(ed.points, ed.points, num_classes=int(sf.opt.num_classes), seg1=ed.seg, seg2=ed.seg,
                             k=k, skip_self=True)
        w, _ = _weights64(idx, dist, ed.radii, 1)
        ed.knn_w, ed.knn_indices = w, idx
        return
    dist, idx = find_knn(ed.points, ed.points, k=k, skip_self=True)
    if dist.dtype == torch.float64:
        w, _ = _weights64(idx, dist, ed.radii, 1)
    else:
        w, _ = _weights(idx, dist, ed.radii, 1)
    ed.knn_w = w
    ed.knn_indices = idx


def update_sfed_knn(sf):
    """Surfel-node KNN, stability test ``any(dist <= radius)`` and
    ``softmax(exp(-dist/radius))`` weights (Semantic-SuPer: class-restricted neighbours under
    ``hard_seg``, Jensen-Shannon weights otherwise)."""
    ed = sf.ED_nodes
    k = int(sf.opt.num_neighbors)
    hard = bool(getattr(sf, "hard_seg", False))
    soft = getattr(sf.opt, "method", "super") == "semantic-super" and not hard
    if hard or soft:
        if hard:
            dist, idx = find_knn(sf.points, ed.points, num_classes=int(sf.opt.num_classes), seg1=sf.seg, seg2=ed.seg, k=k)
        else:
            # float64 distances for the float64 weights
            dist, idx = find_knn(sf.points.to(torch.float64), ed.points, k=k)
        w, st8 = _weights64(idx, dist, ed.radii, 0, stable=sf.isStable,
                            q_conf=sf.seg_conf if soft else None, node_conf=ed.seg_conf if soft else None)
        sf.knn_indices, sf.knn_w, sf.isStable = idx, w, st8.to(torch.bool)
        return
    dist, idx = find_knn(sf.points, ed.points, k=k)
    if dist.dtype == torch.float64:
        w, st8 = _weights64(idx, dist, ed.radii, 0, stable=sf.isStable)
    else:
        w, st8 = _weights(idx, dist, ed.radii, 0, stable=sf.isStable)
    sf.knn_indices = idx
    sf.knn_w = w
    sf.isStable = st8.to(torch.bool)


def update(sf, deform):
    """Apply the solved warp: skin surfel points, blend and normalise surfel normals, translate
    nodes, rotate node normals.  ``deform`` is (J,7) on the LM path
    (``opt.use_derived_gradient``) and (J+1,7) with the global row T_g on the autograd path.
    The state keeps its dtype: float64 tensors (the reference's) are updated in float64, nothing
    is rounded; float32 tensors stay float32."""
    if deform is None:
        return
    lib = _lib.load()
    dev = sf.points.device
    ed = sf.ED_nodes
    sdt = torch.float64 if sf.points.dtype == torch.float64 else torch.float32
    pts, nrm = _as(sf.points, sdt, dev).clone(), _as(sf.norms, sdt, dev).clone()
    epts, enrm = _as(ed.points, sdt, dev).clone(), _as(ed.norms, sdt, dev).clone()
    idx = _as(sf.knn_indices, torch.int32, dev)
    w = _as(sf.knn_w, sdt, dev)
    beta = _as(deform, torch.float64, dev)
    f64 = sdt == torch.float64
    fn = lib.slm_apply_update_f64 if f64 else lib.slm_apply_update
    if beta.shape[0] == epts.shape[0] + 1:
        fn = lib.slm_apply_update_gf_f64 if f64 else lib.slm_apply_update_gf
    elif beta.shape[0] != epts.shape[0]:
        raise ValueError("deform must have J or J+1 rows")
    _lib.check(fn(pts.shape[0], epts.shape[0], idx.shape[1], _dev_ptr(pts),
                  _dev_ptr(nrm), _dev_ptr(idx), _dev_ptr(w), _dev_ptr(epts),
                  _dev_ptr(enrm), _dev_ptr(beta), _stream_ptr(dev)),
               "slm_apply_update")
    sf.points, sf.norms = pts.to(sf.points.dtype), nrm.to(sf.norms.dtype)
    ed.points, ed.norms = epts.to(ed.points.dtype), enrm.to(ed.norms.dtype)
