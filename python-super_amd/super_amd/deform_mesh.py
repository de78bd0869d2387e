"""Host-side mirror of the reference's default per-frame optimiser ``GraphFit``
(``super/deform_mesh.py:10-379``) over libsuper_lm.so.

Same constructor and ``forward(inputs, src, trg, models)`` surface as the reference class,
so ``SuPer.__init__`` / ``SuPer.fusion`` (``super/super.py:20-21,70``) can use it unchanged:

    self.graph_fit = GraphFit(self.opt)
    deform_param = self.graph_fit(inputs, self.sf, sfdata, models)   # (J+1,7) float64

Supported loss flags: ``sf_point_plane``, ``mesh_arap``, ``mesh_rot``, ``mesh_face`` with their
weights, the Semantic-SuPer terms ``sf_soft_seg_point_plane`` / ``sf_hard_seg_point_plane`` /
``sf_bn_morph`` (+ ``sf_bn_morph_weight``, ``num_classes``), the ``max`` clip of the point-plane
term that ``depth_model == "raft_stereo"`` switches on, ``optimizer`` in {"SGD", "Adam"},
``learning_rate``, ``num_optimize_iterations``, and the surfel-correspondence term ``sf_corr``
(+ ``sf_corr_weight``, ``sf_corr_loss_type``): the flow network stays the caller's -- like the reference
(deform_mesh.py:19-23,302-309) ``forward`` calls ``models.optical_flow(src.rgb, inputs[("color",0)])`` once per
frame and hands the (1,2,H,W) flow to the library.  ``sf_corr_match_renderimg`` (flow re-inferred from the
rendered image every iteration) and the (unused) render loss raise ``NotImplementedError``.
The renderer call the reference makes every iteration (deform_mesh.py:294-298) only feeds those two and is
not needed here.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import SlmGfConfig, SlmGfFrame, SlmGfSemantic
from .LM import BoundFrame, _as, _dev_ptr, _stream_ptr


class GraphFit:
    """``GraphFit(opt)`` as in the reference.  ``rank`` / ``world`` (default: the
    ``torch.distributed`` rank and world size when ``shard_surfels=True``) split the surfels of
    one large frame over the GPUs of a node: every rank evaluates its block of surfels, the
    partial gradient / loss sums are all-reduced (RCCL) twice per optimiser iteration and every
    rank takes the same step (BASELINE configs[4])."""

    def __init__(self, opt, max_frames=1, shard_surfels=False, rank=None, world=None, all_reduce=None):
        self.opt = opt
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.SuperLMError("no HIP device visible: super_amd has no CPU fallback")
        for flag in ("sf_corr_match_renderimg", "render_loss"):
            if getattr(opt, flag, False) and (flag == "render_loss" or getattr(opt, "sf_corr", False)):
                raise NotImplementedError(f"super_amd.GraphFit: opt.{flag} is not supported")
        self.valid_margin = 1
        self.optim = opt.optimizer
        self.Niter = opt.num_optimize_iterations
        if self.optim not in ("SGD", "Adam"):
            raise NotImplementedError(f"optimizer {self.optim!r}")
        self.max_frames = max_frames
        cfg = SlmGfConfig()
        cfg.num_iterations = int(self.Niter)
        cfg.optimizer = 0 if self.optim == "SGD" else 1
        cfg.use_data = int(bool(opt.sf_point_plane))
        cfg.use_arap = int(bool(opt.mesh_arap))
        cfg.use_rot = int(bool(opt.mesh_rot))
        cfg.use_face = int(bool(getattr(opt, "mesh_face", False)))
        cfg.max_frames = max_frames
        cfg.w_data = float(getattr(opt, "sf_point_plane_weight", 1.0))
        cfg.w_arap = float(getattr(opt, "mesh_arap_weight", 10.0))
        cfg.w_rot = float(getattr(opt, "mesh_rot_weight", 1.0))
        cfg.w_face = float(getattr(opt, "mesh_face_weight", 1.0))
        cfg.lr = float(opt.learning_rate)
        # Semantic-SuPer (deform_mesh.py:76-99,126-194): soft wins over hard (loss.py:384)
        soft = bool(getattr(opt, "sf_soft_seg_point_plane", False))
        hard = bool(getattr(opt, "sf_hard_seg_point_plane", False))
        cfg.seg_mode = 2 if soft else (1 if hard else 0)
        cfg.use_bn_morph = int(bool(getattr(opt, "sf_bn_morph", False)))
        cfg.w_bn_morph = float(getattr(opt, "sf_bn_morph_weight", 0.1))
        cfg.pp_max = 2e-5 if getattr(opt, "depth_model", None) == "raft_stereo" else 0.0
        if getattr(opt, "sf_corr", False):                 # deform_mesh.py:100-109
            lt = getattr(opt, "sf_corr_loss_type", "point-point")
            if lt not in ("point-point", "point-plane"):
                raise ValueError(f"sf_corr_loss_type {lt!r}")
            cfg.corr_mode = 1 if lt == "point-point" else 2
            cfg.w_corr = float(getattr(opt, "sf_corr_weight", 0.001))
        self.semantic = bool(cfg.seg_mode or cfg.use_bn_morph)
        self.flow = None
        self.edge_counts = None
        self.cfg = cfg
        self.h = C.c_void_p()
        _lib.check(self.lib.slm_gf_create(C.byref(cfg), C.byref(self.h)), "slm_gf_create")
        self._keep = [None] * max_frames
        self.rank, self.world, self._all_reduce = 0, 1, all_reduce
        self.sharded = bool(shard_surfels or world is not None)   # (a world of one rank runs the same protocol)
        if self.sharded:
            import torch.distributed as dist
            if world is None:
                world, rank = dist.get_world_size(), dist.get_rank()
            self.rank, self.world = int(rank), int(world)
            if self._all_reduce is None:
                from .dist import default_collectives
                self._all_reduce = default_collectives()[0]         # sum, in place (RCCL on the GPU box; host-staged under gloo)
            _lib.check(self.lib.slm_gf_set_shard(self.h, self.rank, self.world), "slm_gf_set_shard")

    def __del__(self):
        try:
            if self.h:
                self.lib.slm_gf_destroy(self.h)
        except Exception:
            pass

    def infer_flow(self, models, source_img, target_img):
        """(reference ``deform_mesh.py:19-23``) the caller's flow network; the last element of a list output."""
        flow = models.optical_flow(source_img, target_img)   # x, y
        if isinstance(flow, (list, tuple)):
            flow = flow[-1]
        return flow.detach()

    def _bind(self, slot, inputs, src, trg, models=None, flow=None):
        new_data = trg
        if not hasattr(trg, "valid"):            # not read on this path
            trg = type("T", (), {})()
            trg.points, trg.norms, trg.index_map = new_data.points, new_data.norms, new_data.index_map
            trg.valid = torch.zeros(new_data.index_map.numel(), dtype=torch.bool,
                                    device=new_data.points.device)
        bf = BoundFrame(src, inputs, trg, state=getattr(self.opt, "slm_state_dtype", None))
        dev = bf.device
        sdt = bf.state_dtype       # ED_nodes.knn_w / triangle areas follow the dtype of the state
        ed = src.ED_nodes
        fr = SlmGfFrame()
        fr.base = bf.c
        keep = [bf]
        stable = getattr(src, "isStable", None)
        if stable is not None:
            st8 = _as(stable, torch.uint8, dev)
            keep.append(st8)
            fr.sf_stable = _dev_ptr(st8)
        w = _as(ed.knn_w, sdt, dev)
        keep.append(w)
        fr.ed_knn_w = _dev_ptr(w)
        if self.cfg.use_face:
            tri = _as(ed.triangles, torch.int32, dev)
            area = _as(ed.triangles_areas, sdt, dev)
            keep += [tri, area]
            fr.ed_triangles, fr.ed_triangle_areas = _dev_ptr(tri), _dev_ptr(area)
            fr.n_triangles = int(tri.shape[1])
        _lib.check(self.lib.slm_gf_bind_frame(self.h, slot, C.byref(fr), _stream_ptr(dev)),
                   "slm_gf_bind_frame")
        if self.semantic:
            # src.seg / src.seg_conf (deform_mesh.py:262-264), trg.seg_conf (loss.py:350),
            # inputs[("seg_conf",0)] / inputs[("seg",0)] (deform_mesh.py:136,149)
            sem = SlmGfSemantic()
            nc = int(getattr(self.opt, "num_classes", src.seg_conf.shape[1]))
            sem.num_classes = nc
            seg = _as(src.seg, torch.int32, dev)
            keep.append(seg)
            sem.sf_seg = _dev_ptr(seg)
            if self.cfg.seg_mode:
                sconf = _as(src.seg_conf, torch.float32, dev)
                tconf = _as(new_data.seg_conf, torch.float32, dev)
                if sconf.shape[1] != nc or tconf.shape[1] != nc:
                    raise ValueError("seg_conf must have opt.num_classes columns")
                keep += [sconf, tconf]
                sem.sf_seg_conf, sem.tgt_seg_conf = _dev_ptr(sconf), _dev_ptr(tconf)
            if self.cfg.use_bn_morph:
                iconf = _as(inputs[("seg_conf", 0)][0], torch.float32, dev)
                iseg = inputs[("seg", 0)]
                if iseg.shape[1] > 1:                      # find_edge_region: argmax over channels
                    iseg = torch.argmax(iseg, dim=1, keepdim=True)
                iseg = _as(iseg[0, 0], torch.int32, dev)
                if iconf.shape[0] != nc:
                    raise ValueError('inputs[("seg_conf",0)] must have opt.num_classes channels')
                keep += [iconf, iseg]
                sem.img_seg_conf, sem.img_seg = _dev_ptr(iconf), _dev_ptr(iseg)
            counts = (C.c_int32 * 4)()
            _lib.check(self.lib.slm_gf_bind_semantic(self.h, slot, C.byref(sem), counts, _stream_ptr(dev)),
                       "slm_gf_bind_semantic")
            self.edge_counts = list(counts)[:nc]
        if self.cfg.corr_mode:
            if flow is None:
                if models is None or not hasattr(models, "optical_flow"):
                    raise ValueError("opt.sf_corr needs models.optical_flow (or flow=...)")   # the reference asserts
                flow = self.infer_flow(models, src.rgb, inputs[("color", 0)])
            self.flow = flow
            fl = _as(flow, torch.float32, dev)
            if tuple(fl.shape) != (1, 2, bf.c.H, bf.c.W):
                raise ValueError(f"flow must be (1,2,{bf.c.H},{bf.c.W}), got {tuple(fl.shape)}")
            keep.append(fl)
            _lib.check(self.lib.slm_gf_bind_flow(self.h, slot, _dev_ptr(fl), _stream_ptr(dev)), "slm_gf_bind_flow")
        self._keep[slot] = keep
        return bf

    def forward(self, inputs, src, trg, models=None):
        """(reference ``deform_mesh.py:232-247``) returns deform_verts (J+1,7) float64."""
        if getattr(self.opt, "deform_udpate_method", "super_edg") != "super_edg":
            raise NotImplementedError("only deform_udpate_method == 'super_edg'")
        bf = self._bind(0, inputs, src, trg, models)
        st = _stream_ptr(bf.device)
        if self.sharded:
            part = torch.empty((bf.J + 1) * 7 + _lib.GF_NTERMS, dtype=torch.float64, device=bf.device)
            for _ in range(int(self.Niter)):
                self.eval_morph()
                if self.cfg.use_bn_morph:
                    self.exchange_partial(part)      # global kept count before the back-propagation
                self.eval_losses()
                self.exchange_partial(part)          # gradient + loss terms
                self.step()
        else:
            _lib.check(self.lib.slm_gf_run(self.h, 1, st), "slm_gf_run")
        out = torch.empty((bf.J + 1, 7), dtype=torch.float64, device=bf.device)
        _lib.check(self.lib.slm_gf_get_deform(self.h, 0, _dev_ptr(out), st), "slm_gf_get_deform")
        return out

    __call__ = forward

    # ---- stepwise evaluation (surfel-sharded frames; also usable with world == 1) -------------
    def _st(self):
        return _stream_ptr(self._keep[0][0].device)

    def bind(self, inputs, src, trg, models=None, flow=None):
        return self._bind(0, inputs, src, trg, models, flow)

    def eval_morph(self):
        _lib.check(self.lib.slm_gf_eval_morph(self.h, 1, self._st()), "slm_gf_eval_morph")

    def eval_losses(self):
        _lib.check(self.lib.slm_gf_eval_losses(self.h, 1, self._st()), "slm_gf_eval_losses")

    def step(self):
        _lib.check(self.lib.slm_gf_step(self.h, 1, self._st()), "slm_gf_step")

    def get_partial(self, out):
        _lib.check(self.lib.slm_gf_get_partial(self.h, 0, _dev_ptr(out), self._st()), "slm_gf_get_partial")
        return out

    def set_partial(self, buf):
        _lib.check(self.lib.slm_gf_set_partial(self.h, 0, _dev_ptr(buf), self._st()), "slm_gf_set_partial")

    def exchange_partial(self, buf):
        """partial [(J+1)*7 gradient | GF_NTERMS terms] -> sum over the ranks -> back into the slot."""
        self.get_partial(buf)
        self._all_reduce(buf)
        self.set_partial(buf)

    def deform_verts(self):
        bf = self._keep[0][0]
        out = torch.empty((bf.J + 1, 7), dtype=torch.float64, device=bf.device)
        _lib.check(self.lib.slm_gf_get_deform(self.h, 0, _dev_ptr(out), self._st()), "slm_gf_get_deform")
        return out

    def loss_and_grad(self, inputs, src, trg, deform_verts, models=None, flow=None):
        """One evaluation of ``deform_source`` + ``get_losses`` + backward at ``deform_verts``:
        returns (dict of weighted loss terms, matched count, gradient (J+1,7) with the global
        row divided by J)."""
        bf = self._bind(0, inputs, src, trg, models, flow)
        st = _stream_ptr(bf.device)
        dv = _as(deform_verts, torch.float64, bf.device)
        terms = torch.zeros(_lib.GF_NTERMS, dtype=torch.float64, device=bf.device)
        grad = torch.zeros((bf.J + 1, 7), dtype=torch.float64, device=bf.device)
        _lib.check(self.lib.slm_gf_loss_grad(self.h, 0, _dev_ptr(dv), _dev_ptr(terms), _dev_ptr(grad), st),
                   "slm_gf_loss_grad")
        t = terms.cpu().tolist()
        d = dict(face_losses=t[0], arap_loss=t[1], rot_loss=t[2], point_plane_loss=t[3])
        if self.cfg.use_bn_morph and t[7] != 0.0:       # the reference only adds the key when a class contributes
            d["sf_bn_morph_loss"] = t[5]
        self.last_bn_morph_kept = int(t[6])
        if self.cfg.corr_mode:
            d["corr_loss"] = t[8]
            self.last_corr_kept = int(t[9])
        return d, int(t[4]), grad

    def edge_points(self, class_id):
        """Boundary pixels (x,y) of ``class_id`` extracted at the last bind (``self.edge_pts`` of the
        reference, deform_mesh.py:145-165), float32 (E,2) on the device."""
        n = self.edge_counts[class_id]
        dev = self._keep[0][0].device
        out = torch.empty((n, 2), dtype=torch.float32, device=dev)
        _lib.check(self.lib.slm_gf_get_edge_points(self.h, 0, class_id, _dev_ptr(out), n, _stream_ptr(dev)),
                   "slm_gf_get_edge_points")
        return out
