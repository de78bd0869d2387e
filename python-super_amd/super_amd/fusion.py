"""Host-side mirrors of the reference's surfel fusion methods over
libsuper_lm.so:

* :func:`fuseInputData`                    <- ``Surfels.fuseInputData``                   (``super/nodes.py:268-541``)
* :func:`prepareStableIndexNSwapAllModel`  <- ``Surfels.prepareStableIndexNSwapAllModel`` (``super/nodes.py:543-585``)

Both take the reference's ``sf`` object (anything with the same attributes) and can be bound onto the
reference class: ``Surfels.fuseInputData = super_amd.fusion.fuseInputData``.  Supported:
``opt.method == "super"`` and ``"semantic-super"`` (segmentation fields ``seg`` / ``seg_conf`` /
``dist2edge`` fused, appended and compacted; Jensen-Shannon skinning weights; ``hard_seg``
class-restricted neighbours, ``super/nodes.py:314-316,348-353,467-509``), with or without tracked evaluation points
(``sf.track_id``: re-pointed when their surfel is absorbed, dropped (-2) when it is deleted, kept
alive and renumbered by the swap, ``super/nodes.py:440-456,559-590``); the logging / rendering calls
at the end of the reference's swap are not part of the mirror.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import SlmFuseConfig, SlmFuseSemantic, SlmNewFrame, SlmSurfelModel
from .LM import _as, _dev_ptr, _stream_ptr

_ctx = {}


def _context(lib, H, W, cap, dev):
    key = (H, W, str(dev))
    h, have = _ctx.get(key, (None, 0))
    if h is None or have < cap:
        if h is not None:
            lib.slm_fuse_destroy(h)
        h = C.c_void_p()
        cap = int(cap * 1.25) + 1024
        _lib.check(lib.slm_fuse_create(H, W, cap, C.byref(h)), "slm_fuse_create")
        _ctx[key] = (h, cap)
    return _ctx[key]


def _config(sf, inputs):
    o = sf.opt
    if getattr(o, "method", "super") not in ("super", "semantic-super"):
        raise NotImplementedError("super_amd.fusion: opt.method must be 'super' or 'semantic-super'")
    K = inputs["K"][0].detach().cpu().float()
    c = SlmFuseConfig()
    c.H, c.W = int(o.height), int(o.width)
    c.merge_new = int(not o.disable_merging_new_surfels)
    c.merge_exist = int(not o.disable_merging_exist_surfels)
    c.add_new = int(not o.disable_adding_new_surfels)
    c.remove_unstable = int(not o.disable_removing_unstable_surfels)
    c.phase_test = int(o.phase == "test")
    c.th_time_steps = int(o.th_time_steps)
    c.th_dist, c.th_cosine_ang = float(o.th_dist), float(o.th_cosine_ang)
    c.fx, c.fy, c.cx, c.cy = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
    return c


class _Model:
    """Device buffers with spare rows around the surfel arrays of ``sf``."""

    FIELDS = (("points", torch.float64, 3), ("norms", torch.float64, 3), ("colors", torch.float32, 3),
              ("radii", torch.float64, 0), ("confs", torch.float32, 0), ("time_stamp", torch.float32, 0),
              ("isStable", torch.uint8, 0), ("knn_indices", torch.int32, 4), ("knn_w", torch.float64, 4),
              ("projdata", torch.float32, 2))
    SEG_DTYPES = {"seg": torch.long, "seg_conf": torch.float64, "dist2edge": torch.float64}

    def __init__(self, sf, cap, dev):
        n = int(sf.points.shape[0])
        self.n, self.cap, self.dev = n, cap, dev
        self.buf = {}
        # opt.num_neighbors (K-generic like the reference's find_knn / skinning weights, super/nodes.py:170-191,466-509)
        K = int(sf.knn_indices.shape[1]) if getattr(sf, "knn_indices", None) is not None and sf.knn_indices.dim() == 2 \
            else int(getattr(sf.opt, "num_neighbors", 4))
        if not 1 <= K <= 8:
            raise NotImplementedError("super_amd.fusion: opt.num_neighbors must be in 1..8")
        for name, dt, width in self.FIELDS:
            width = K if name in ("knn_indices", "knn_w") else width
            shape = (cap, width) if width else (cap,)
            b = torch.zeros(shape, dtype=dt, device=dev)
            src = getattr(sf, name, None)
            if src is not None and src.shape[0] == n:
                b[:n] = _as(src, dt, dev)
            self.buf[name] = b
        ed = sf.ED_nodes
        self.ed_points = _as(ed.points, torch.float64, dev)
        self.ed_radii = _as(ed.radii, torch.float64, dev)
        m = SlmSurfelModel()
        m.n, m.cap = n, cap
        for cname, name in (("points", "points"), ("norms", "norms"), ("colors", "colors"), ("radii", "radii"),
                            ("confs", "confs"), ("time_stamp", "time_stamp"), ("is_stable", "isStable"),
                            ("knn_idx", "knn_indices"), ("knn_w", "knn_w"), ("projdata", "projdata")):
            setattr(m, cname, _dev_ptr(self.buf[name]))
        m.J = int(self.ed_points.shape[0])
        m.K = K
        m.ed_points, m.ed_radii = _dev_ptr(self.ed_points), _dev_ptr(self.ed_radii)
        self.merged_into = None
        if hasattr(sf, "track_pts") or getattr(sf, "evaluate_tracking", False):
            self.merged_into = torch.full((cap,), -1, dtype=torch.int32, device=dev)
            m.merged_into = _dev_ptr(self.merged_into)
        self.c = m
        self.sem = None
        if hasattr(sf, "seg"):
            self._bind_segmentation(sf, n, cap, dev)

    def _bind_segmentation(self, sf, n, cap, dev):
        """sf.seg / seg_conf / dist2edge with spare rows + the ED nodes' class fields."""
        o = sf.opt
        C_ = int(sf.seg_conf.shape[1])
        if C_ > _lib.SLM_MAX_CLASSES:
            raise NotImplementedError(f"super_amd.fusion: at most {_lib.SLM_MAX_CLASSES} classes")
        if not hasattr(sf, "dist2edge"):
            raise ValueError("super_amd.fusion: sf.seg without sf.dist2edge (the reference keeps both, nodes.py:575)")
        self.buf["seg"] = torch.zeros(cap, dtype=torch.int32, device=dev)
        self.buf["seg_conf"] = torch.zeros((cap, C_), dtype=torch.float64, device=dev)
        self.buf["dist2edge"] = torch.zeros(cap, dtype=torch.float64, device=dev)
        self.buf["seg"][:n] = _as(sf.seg, torch.int32, dev)
        self.buf["seg_conf"][:n] = _as(sf.seg_conf, torch.float64, dev)
        self.buf["dist2edge"][:n] = _as(sf.dist2edge, torch.float64, dev)
        s = SlmFuseSemantic()
        s.num_classes = C_
        hard = bool(getattr(sf, "hard_seg", False))
        s.soft_weights = int(getattr(o, "method", "super") == "semantic-super")
        s.hard_seg = int(hard)
        s.merge_same_class = int(hard or getattr(o, "data", "") == "superv1")
        s.seg, s.seg_conf, s.dist2edge = (_dev_ptr(self.buf[k]) for k in ("seg", "seg_conf", "dist2edge"))
        ed = sf.ED_nodes
        if s.soft_weights:
            self.ed_seg_conf = _as(ed.seg_conf, torch.float64, dev)
            s.ed_seg_conf = _dev_ptr(self.ed_seg_conf)
        if hard:
            self.ed_seg = _as(ed.seg, torch.int32, dev)
            s.ed_seg = _dev_ptr(self.ed_seg)
        self.sem = s

    def bind_frame_segmentation(self, sfdata, dev):
        self.new_seg = _as(sfdata.seg, torch.int32, dev)
        self.new_seg_conf = _as(sfdata.seg_conf, torch.float64, dev)
        self.new_dist2edge = _as(sfdata.dist2edge, torch.float64, dev)
        self.sem.new_seg, self.sem.new_seg_conf = _dev_ptr(self.new_seg), _dev_ptr(self.new_seg_conf)
        self.sem.new_dist2edge = _dev_ptr(self.new_dist2edge)

    def write_back(self, sf):
        n = int(self.c.n)
        ref = {"points": torch.float64, "norms": torch.float64, "colors": torch.float32, "radii": torch.float64,
               "confs": torch.float32, "time_stamp": torch.float32, "isStable": torch.bool,
               "knn_indices": torch.long, "knn_w": torch.float64, "projdata": torch.float32}
        for name, dt in ref.items():
            setattr(sf, name, self.buf[name][:n].to(dt))
        if self.sem is not None:
            for name, dt in self.SEG_DTYPES.items():
                setattr(sf, name, self.buf[name][:n].to(dt))


def fuseInputData(sf, inputs, sfdata):
    lib = _lib.load()
    cfg = _config(sf, inputs)
    dev = sf.points.device if sf.points.is_cuda else torch.device("cuda", torch.cuda.current_device())
    n, T = int(sf.points.shape[0]), int(sfdata.points.shape[0])
    cap = n + T
    h, _ = _context(lib, cfg.H, cfg.W, cap, dev)
    model = _Model(sf, cap, dev)
    fr = SlmNewFrame()
    fr.T, fr.time = T, int(sfdata.time)
    keep = dict(points=_as(sfdata.points, torch.float64, dev), norms=_as(sfdata.norms, torch.float64, dev),
                colors=_as(sfdata.colors, torch.float32, dev), radii=_as(sfdata.radii, torch.float64, dev),
                confs=_as(sfdata.confs, torch.float32, dev), valid=_as(sfdata.valid, torch.uint8, dev),
                index_map=_as(sfdata.index_map, torch.int32, dev))
    for k, v in keep.items():
        setattr(fr, k, _dev_ptr(v))
    sf.time = sfdata.time
    stable_before = model.buf["isStable"][:n].clone()
    if model.sem is not None:
        model.bind_frame_segmentation(sfdata, dev)
    _lib.check(lib.slm_fuse_bind_semantic(h, C.byref(model.sem) if model.sem is not None else None),
               "slm_fuse_bind_semantic")
    _lib.check(lib.slm_fuse_input_data(h, C.byref(cfg), C.byref(model.c), C.byref(fr), _stream_ptr(dev)),
               "slm_fuse_input_data")
    model.write_back(sf)
    if model.merged_into is not None and hasattr(sf, "track_id") and cfg.merge_exist:
        # tracked ids follow the surfel that absorbed theirs; ids of deleted surfels become -2
        tid = sf.track_id.clone()
        live = tid >= 0
        if bool(live.any()):
            t = tid[live].to(torch.long)
            into = model.merged_into[:n].to(torch.long)[t]
            t = torch.where(into >= 0, into, t)
            gone = stable_before.to(torch.bool)[t] & ~sf.isStable[t]
            t = torch.where(gone, torch.full_like(t, -2), t)
            tid[live] = t.to(tid.dtype)
        sf.track_id = tid


def prepareStableIndexNSwapAllModel(sf, inputs, sfdata):
    lib = _lib.load()
    cfg = _config(sf, inputs)
    dev = sf.points.device if sf.points.is_cuda else torch.device("cuda", torch.cuda.current_device())
    n = int(sf.points.shape[0])
    h, _ = _context(lib, cfg.H, cfg.W, n, dev)
    model = _Model(sf, n, dev)
    tracking = bool(getattr(sf, "evaluate_tracking", False)) and hasattr(sf, "track_id")
    keep, new_index, n_keep = None, None, 0
    if tracking and cfg.remove_unstable:
        keep = sf.track_id[sf.track_id >= 0].to(device=dev, dtype=torch.int32).contiguous()
        n_keep = int(keep.numel())
        new_index = torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    _lib.check(lib.slm_fuse_bind_semantic(h, C.byref(model.sem) if model.sem is not None else None),
               "slm_fuse_bind_semantic")
    _lib.check(lib.slm_fuse_swap_stable(h, C.byref(cfg), C.byref(model.c), int(inputs["time"]),
                                        _dev_ptr(keep) if n_keep else None, n_keep,
                                        _dev_ptr(new_index) if new_index is not None else None, _stream_ptr(dev)),
               "slm_fuse_swap_stable")
    model.write_back(sf)
    if tracking:
        tid = sf.track_id.clone()
        live = tid >= 0
        if cfg.remove_unstable and bool(live.any()):
            tid[live] = new_index.to(torch.long)[tid[live].to(torch.long)].to(tid.dtype)
        # ids whose surfel is unstable are given up (nodes.py:585-590)
        live = tid >= 0
        if bool(live.any()):
            bad = ~sf.isStable[tid[live].to(torch.long)]
            t = tid[live]
            t[bad] = -2
            tid[live] = t
        sf.track_id = tid
    sf.surfel_num = int(sf.isStable.count_nonzero())
    if tracking and hasattr(sf, "update_track_pts"):
        name = inputs["filename"][0]
        if int((sf.track_id >= 0).count_nonzero()) > 0:
            sf.update_track_pts(sfdata, name)
        if int((sf.track_id == -1).count_nonzero()) > 0:
            sf.init_track_pts(sfdata, name)
