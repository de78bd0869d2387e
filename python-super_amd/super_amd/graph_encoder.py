"""Host-side mirror of the reference's ``DirectDeformGraph`` (``super/graph_encoder.py:70-195``, grid
mesh) over libsuper_lm.so: the ED graph built from the first frame.

    mesh_encoder = DirectDeformGraph(opt)
    sfdata.ED_nodes = mesh_encoder(inputs, sfdata)        # super/super.py:52

Returns an attribute bag with the reference's field names: ``points``, ``norms`` (J,3) float64,
``radii`` (J,), ``edge_index`` (2,E) int64, ``edges_lens`` (E,), ``triangles`` (3,F) int64,
``triangles_areas`` (F,), ``num``, ``param_num``.  Supported: the default ``grid_mesh`` construction for
``opt.method == "super"`` and ``"semantic-super"`` (``seg`` / ``seg_conf`` of the nodes when the frame carries
``data.seg``; ``opt.hard_seg`` with ``opt.mesh_face`` drops edges / triangles across a class boundary,
``super/graph_encoder.py:134-151,190-192``); the ``ball_pivoting`` (open3d) and ``knn`` variants are not
reachable from ``forward`` in the reference either.
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import torch

from . import _lib
from ._lib import SlmGraphOutputs
from .LM import _as, _dev_ptr, _stream_ptr


class DirectDeformGraph:
    def __init__(self, opt):
        self.opt = opt
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.SuperLMError("no HIP device visible: super_amd has no CPU fallback")
        if getattr(opt, "method", "super") not in ("super", "semantic-super"):
            raise NotImplementedError("super_amd.DirectDeformGraph: opt.method must be 'super' or 'semantic-super'")

    def forward(self, inputs, data):
        o = self.opt
        H, W, step = int(o.height), int(o.width), int(o.mesh_step_size)
        dev = data.points.device if data.points.is_cuda else torch.device("cuda", torch.cuda.current_device())
        valid = _as(data.valid.view(-1), torch.uint8, dev)
        imap = _as(data.index_map, torch.int32, dev)
        pts, nrm = _as(data.points, torch.float64, dev), _as(data.norms, torch.float64, dev)
        cap = ((H - 1 + step - 1) // step) * ((W - 1 + step - 1) // step)
        f64 = torch.float64
        buf = dict(points=torch.empty((cap, 3), dtype=f64, device=dev), norms=torch.empty((cap, 3), dtype=f64, device=dev),
                   radii=torch.empty(cap, dtype=f64, device=dev),
                   edge_index=torch.empty((2, 4 * cap), dtype=torch.int32, device=dev),
                   edges_lens=torch.empty(4 * cap, dtype=f64, device=dev),
                   triangles=torch.empty((3, 2 * cap), dtype=torch.int32, device=dev),
                   triangles_areas=torch.empty(2 * cap, dtype=f64, device=dev))
        out = SlmGraphOutputs()
        out.cap_nodes = cap
        for k, v in buf.items():
            setattr(out, k, _dev_ptr(v))
        counts = (C.c_int32 * 3)()
        seg = seg_conf = None
        if hasattr(data, "seg"):
            conf = _as(data.seg_conf, f64, dev)
            C_ = int(conf.shape[1])
            if C_ > _lib.SLM_MAX_CLASSES:
                raise NotImplementedError(f"super_amd.DirectDeformGraph: at most {_lib.SLM_MAX_CLASSES} classes")
            seg = torch.empty(cap, dtype=torch.int32, device=dev)
            seg_conf = torch.empty((cap, C_), dtype=f64, device=dev)
            prune = bool(getattr(o, "hard_seg", False)) and bool(getattr(o, "mesh_face", False))
            _lib.check(self.lib.slm_graph_init_semantic(H, W, step, _dev_ptr(valid), _dev_ptr(imap), _dev_ptr(pts),
                                                        _dev_ptr(nrm), C_, _dev_ptr(conf), int(prune), C.byref(out),
                                                        _dev_ptr(seg), _dev_ptr(seg_conf), counts, _stream_ptr(dev)),
                       "slm_graph_init_semantic")
        else:
            _lib.check(self.lib.slm_graph_init(H, W, step, _dev_ptr(valid), _dev_ptr(imap), _dev_ptr(pts), _dev_ptr(nrm),
                                               C.byref(out), counts, _stream_ptr(dev)), "slm_graph_init")
        J, E, F = int(counts[0]), int(counts[1]), int(counts[2])
        graph = SimpleNamespace(points=buf["points"][:J], norms=buf["norms"][:J], radii=buf["radii"][:J],
                                edge_index=buf["edge_index"][:, :E].to(torch.long), edges_lens=buf["edges_lens"][:E],
                                triangles=buf["triangles"][:, :F].to(torch.long),
                                triangles_areas=buf["triangles_areas"][:F], num=J, param_num=7 * J)
        if getattr(o, "method", "super") == "semantic-super" and seg is not None:      # graph_encoder.py:190-192
            graph.seg, graph.seg_conf = seg[:J].to(torch.long), seg_conf[:J]
        return graph

    __call__ = forward
