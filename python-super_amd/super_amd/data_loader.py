"""Host-side mirror of the reference's ``depth_preprocessing`` (``utils/data_loader.py:333-523``,
SURVEY.md 8(f) row f2) over libsuper_lm.so: depth map -> the per-frame target ``sfdata`` that
``LM_Solver.LM`` / ``GraphFit`` consume.

    data, inputs = depth_preprocessing(opt, models, inputs)

reads ``inputs[("depth",0)]`` (1,1,H,W), ``inputs["inv_K"]`` / ``inputs["K"]`` (1,4,4),
``inputs[("color",0)]`` (1,3,H,W), ``inputs["divterm"]``, optionally ``inputs[("seg",0)]`` /
``inputs[("seg_conf",0)]`` and ``opt.{height,width,data,load_depth,depth_width_range,depth_model,
dilate_invalid_kernel,normal_model,del_seg_classes,num_classes,phase}``, and returns an attribute bag
with the reference's field names and dtypes (points / norms float64, radii float64, confs float32,
index_map int64, valid bool, ...).  Like the reference it NaNs ``inputs[("depth",0)]`` /
``inputs[("disp",0)]`` at the invalid pixels and stores ``inputs["valid_map"]`` (superv1).
With ``opt.disable_ssim_conf == False`` (the CLI default) the stereo confidence is computed on the device
too -- the image warped through ``K @ inputs["stereo_T"]`` at the back-projected depth and its 7x7-window
SSIM against itself (``skimage.metrics.structural_similarity`` semantics of the 0.19 line, see
``oracle/depth_oracle.py::skimage_ssim_full``) -- stored in ``inputs[("disp_conf",0)]`` and blended into
``data.confs`` (``utils/data_loader.py:359-373,477-479``).
``opt.load_valid_mask`` (superv1) reads the mask image like the reference (Pillow instead of OpenCV; an
``inputs["valid_mask"]`` (H,W) tensor takes precedence).  The bilateral ``pcd2norm`` branch of the reference is
unreachable (``inputs["pcd"]`` always exists by then) and is not built.
"""
from __future__ import annotations

import ctypes as C
from types import SimpleNamespace

import torch

from . import _lib
from ._lib import SlmDepthConfig, SlmDepthInputs, SlmDepthOutputs
from .LM import _as, _dev_ptr, _stream_ptr

_ctx = {}


def _context(lib, H, W, dev):
    key = (H, W, str(dev))
    if key not in _ctx:
        h = C.c_void_p()
        _lib.check(lib.slm_depth_create(H, W, C.byref(h)), "slm_depth_create")
        _ctx[key] = h
    return _ctx[key]


def depth_preprocessing(opt, models, inputs, return_valid_map=False):
    lib = _lib.load()
    if not torch.cuda.is_available():
        raise _lib.SuperLMError("no HIP device visible: super_amd has no CPU fallback")
    use_ssim = hasattr(opt, "disable_ssim_conf") and not opt.disable_ssim_conf
    depth_t = inputs[("depth", 0)]
    dev = depth_t.device if depth_t.is_cuda else torch.device("cuda", torch.cuda.current_device())
    H, W = int(opt.height), int(opt.width)
    f32 = torch.float32
    depth = _as(depth_t[0, 0], f32, dev)
    color = _as(inputs[("color", 0)][0], f32, dev)
    K = inputs["K"][0].detach().cpu().float()
    iK = inputs["inv_K"][0].detach().cpu().float()
    cfg = SlmDepthConfig()
    cfg.H, cfg.W = H, W
    cfg.data_mode = {"superv1": 0, "superv2": 1}[opt.data]
    cfg.raft_stereo = int(getattr(opt, "depth_model", "") == "raft_stereo")
    cfg.dilate_invalid_kernel = int(getattr(opt, "dilate_invalid_kernel", 0))
    cfg.load_depth = int(bool(getattr(opt, "load_depth", False)))
    cfg.normal_model = {"naive": 0, "8neighbors": 1}[opt.normal_model]
    dels = list(getattr(opt, "del_seg_classes", []) or [])
    if len(dels) > 3:
        raise ValueError("at most 3 del_seg_classes")
    cfg.n_del_classes = len(dels)
    for i, c in enumerate(dels):
        cfg.del_classes[i] = int(c)
    rng = getattr(opt, "depth_width_range", (0.0, 1.0))
    cfg.depth_width_range[0], cfg.depth_width_range[1] = float(rng[0]), float(rng[1])
    for i in range(3):
        for j in range(3):
            cfg.inv_K[3 * i + j] = float(iK[i, j])
    cfg.fx, cfg.fy, cfg.cx, cfg.cy = float(K[0, 0]), float(K[1, 1]), float(K[0, 2]), float(K[1, 2])
    cfg.divterm = float(inputs["divterm"])
    if use_ssim:
        if "stereo_T" not in inputs:
            raise KeyError('depth_preprocessing: inputs["stereo_T"] is needed for the SSIM confidence '
                           "(opt.disable_ssim_conf is False)")
        P = torch.matmul(inputs["K"].detach().cpu().float(), inputs["stereo_T"].detach().cpu().float())[0, :3, :]
        cfg.use_ssim_conf = 1
        for i in range(3):
            for j in range(4):
                cfg.stereo_P[4 * i + j] = float(P[i, j])
    keep = [depth, color]
    inp = SlmDepthInputs()
    inp.depth, inp.color = _dev_ptr(depth), _dev_ptr(color)
    if getattr(opt, "load_valid_mask", False) and opt.data == "superv1":
        if "valid_mask" in inputs:
            vm = _as(inputs["valid_mask"], torch.uint8, dev)
        else:
            # the reference reads <data_dir>/<valid_mask_dir>/<filename>-left.png as a grey image
            # (cv2.imread(..., 0), utils/data_loader.py:376-383); non-zero = valid
            import os
            import numpy as np
            from PIL import Image
            path = os.path.join(opt.data_dir, opt.valid_mask_dir, inputs["filename"][0] + "-left.png")
            if not os.path.exists(path):
                raise FileNotFoundError(f"depth_preprocessing: opt.load_valid_mask is set but {path} does not exist")
            grey = np.asarray(Image.open(path).convert("L"))
            if grey.shape != (H, W):
                raise ValueError(f"depth_preprocessing: valid mask {path} is {grey.shape}, expected {(H, W)}")
            vm = torch.from_numpy((grey != 0).astype(np.uint8)).to(dev)
        keep.append(vm)
        inp.valid_mask = _dev_ptr(vm)
    has_seg = ("seg", 0) in inputs
    C_ = 0
    if has_seg:
        seg = _as(inputs[("seg", 0)][0, 0], torch.int32, dev)
        sconf = _as(inputs[("seg_conf", 0)][0], f32, dev)
        C_ = int(sconf.shape[0])
        cfg.num_classes = int(getattr(opt, "num_classes", C_))
        if cfg.num_classes != C_:
            raise ValueError('inputs[("seg_conf",0)] must have opt.num_classes channels')
        keep += [seg, sconf]
        inp.seg, inp.seg_conf = _dev_ptr(seg), _dev_ptr(sconf)
    n = H * W
    o = dict(points=torch.empty((n, 3), dtype=f32, device=dev), norms=torch.empty((n, 3), dtype=f32, device=dev),
             colors=torch.empty((n, 3), dtype=f32, device=dev), radii=torch.empty(n, dtype=torch.float64, device=dev),
             confs=torch.empty(n, dtype=f32, device=dev), index_map=torch.empty((H, W), dtype=torch.int32, device=dev),
             valid=torch.empty(n, dtype=torch.uint8, device=dev), inval=torch.empty(n, dtype=torch.uint8, device=dev))
    if use_ssim:
        o["disp_conf"] = torch.empty((H, W), dtype=f32, device=dev)
    if has_seg:
        o.update(seg=torch.empty(n, dtype=torch.int32, device=dev),
                 seg_conf=torch.empty((n, C_), dtype=torch.float64, device=dev),
                 dist2edge=torch.empty(n, dtype=torch.float64, device=dev))
    out = SlmDepthOutputs()
    for k, v in o.items():
        setattr(out, k, _dev_ptr(v))
    T = C.c_int32(0)
    _lib.check(lib.slm_depth_preprocess(_context(lib, H, W, dev), C.byref(cfg), C.byref(inp), C.byref(out),
                                        C.byref(T), _stream_ptr(dev)), "slm_depth_preprocess")
    T = T.value
    valid = o["valid"].to(torch.bool)
    inval = o["inval"].to(torch.bool).view(H, W)
    data = SimpleNamespace(points=o["points"][:T].to(torch.float64), norms=o["norms"][:T].to(torch.float64),
                           colors=o["colors"][:T], radii=o["radii"][:T], confs=o["confs"][:T], valid=valid,
                           index_map=o["index_map"].to(torch.long), valid_map=valid.view(H, W))
    if getattr(opt, "phase", "test") != "train" and "filename" in inputs:
        data.time = int(inputs["filename"][0])
    if has_seg:
        data.seg = o["seg"][:T].to(torch.long)
        data.seg_conf = o["seg_conf"][:T]
        data.dist2edge = o["dist2edge"][:T]
    # the reference's side effects on `inputs`
    if use_ssim:
        inputs[("disp_conf", 0)] = o["disp_conf"]
    nan = float("nan")
    if depth_t.is_floating_point():
        depth_t[0, 0][inval.to(depth_t.device)] = nan
    if ("disp", 0) in inputs and inputs[("disp", 0)].is_floating_point():
        inputs[("disp", 0)][0, 0][inval.to(inputs[("disp", 0)].device)] = nan
    if opt.data == "superv1":
        inputs["valid_map"] = ~inval[None, None]
    if not return_valid_map:
        return data, inputs
    return data, inputs, ~inval[None, None]
