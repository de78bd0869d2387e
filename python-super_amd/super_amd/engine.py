"""Thin object wrapper over the C ABI for callers that already hold ABI-layout device
tensors (float32 / int32, contiguous): the benchmark and multi-frame drivers.  The
reference-compatible surface lives in :mod:`super_amd.LM` / :mod:`super_amd.nodes`."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from ._lib import SlmConfig, SlmFrame, SlmIterRecord


@dataclass
class DeviceFrame:
    """One frame pair resident in HBM in the C-ABI layout."""
    H: int
    W: int
    K: np.ndarray                   # (4,4) float32 intrinsics (host)
    sf_points: torch.Tensor         # (N,3) f32 | f64 (model state: one dtype for all six)
    sf_norms: torch.Tensor          # (N,3)
    sf_knn_idx: torch.Tensor        # (N,4) i32
    sf_knn_w: torch.Tensor          # (N,4)
    ed_points: torch.Tensor         # (J,3)
    ed_norms: torch.Tensor          # (J,3)
    ed_knn_idx: torch.Tensor        # (J,K_ED) i32
    tgt_points: torch.Tensor        # (T,3) f32
    tgt_norms: torch.Tensor         # (T,3) f32
    index_map: torch.Tensor         # (H,W) i32
    tgt_valid: torch.Tensor         # (H*W,) u8

    @staticmethod
    def from_scene(sc, device, state_f64=False) -> "DeviceFrame":
        """``state_f64``: hold the model state (surfel / node positions and normals, skinning
        weights) in float64 like the reference; default float32 (BASELINE's fp32 configs)."""
        t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a)).to(device=device, dtype=dt)
        f32, i32 = torch.float32, torch.int32
        sd = torch.float64 if state_f64 else f32
        return DeviceFrame(H=sc.H, W=sc.W, K=sc.K, sf_points=t(sc.sf_points, sd),
                           sf_norms=t(sc.sf_norms, sd), sf_knn_idx=t(sc.sf_knn_idx, i32),
                           sf_knn_w=t(sc.sf_knn_w, sd), ed_points=t(sc.ed_points, sd),
                           ed_norms=t(sc.ed_norms, sd), ed_knn_idx=t(sc.ed_knn_idx, i32),
                           tgt_points=t(sc.tgt_points, f32), tgt_norms=t(sc.tgt_norms, f32),
                           index_map=t(sc.index_map, i32), tgt_valid=t(sc.valid, torch.uint8))

    @property
    def N(self):
        return int(self.sf_points.shape[0])

    @property
    def J(self):
        return int(self.ed_points.shape[0])

    def c_struct(self) -> SlmFrame:
        fr = SlmFrame()
        fr.N, fr.J, fr.T = self.N, self.J, int(self.tgt_points.shape[0])
        fr.H, fr.W = int(self.H), int(self.W)
        fr.K, fr.K_ED = int(self.sf_knn_idx.shape[1]), int(self.ed_knn_idx.shape[1])
        fr.fx, fr.fy, fr.cx, fr.cy = (float(self.K[0, 0]), float(self.K[1, 1]),
                                      float(self.K[0, 2]), float(self.K[1, 2]))
        for name in ("sf_points", "sf_knn_idx", "sf_knn_w", "ed_points", "ed_knn_idx",
                     "tgt_points", "tgt_norms", "index_map", "tgt_valid"):
            t = getattr(self, name)
            assert t.is_cuda and t.is_contiguous()
            setattr(fr, name, t.data_ptr())
        fr.state_f64 = 1 if self.sf_points.dtype == torch.float64 else 0
        assert self.sf_knn_w.dtype == self.sf_points.dtype == self.ed_points.dtype
        return fr


class Engine:
    """A solver handle with ``max_frames`` slots on one device."""

    def __init__(self, device, max_frames=1, num_iterations=10, phase_test=True, use_data=True,
                 use_arap=True, use_rot=True, w_data=1.0, w_arap=10.0, w_rot=1.0, u0=10.0, v=7.5,
                 minimal_loss0=1e10, data_path=0, solver_path=0):
        self.lib = _lib.load()
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        cfg = SlmConfig(num_iterations=num_iterations, phase_test=int(phase_test),
                        use_data=int(use_data), use_arap=int(use_arap), use_rot=int(use_rot),
                        max_frames=max_frames, data_path=data_path, solver_path=solver_path,
                        w_data=w_data, w_arap=w_arap, w_rot=w_rot, u0=u0,
                        v=v, minimal_loss0=minimal_loss0)
        self.cfg = cfg
        self.h = C.c_void_p()
        _lib.check(self.lib.slm_create(C.byref(cfg), C.byref(self.h)), "slm_create")
        self._frames = [None] * max_frames

    def close(self):
        if self.h:
            self.lib.slm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    def bind(self, slot: int, frame: DeviceFrame):
        cs = frame.c_struct()
        _lib.check(self.lib.slm_bind_frame(self.h, slot, C.byref(cs), self.stream), "slm_bind_frame")
        self._frames[slot] = frame

    def bind_batch(self, frames, first: int = 0):
        """Binds ``frames`` to slots ``first ..`` concurrently (``slm_bind_frames``: one host thread / stream per
        frame inside the library, forked from and joined into the current stream)."""
        arr = (_lib.SlmFrame * len(frames))(*[f.c_struct() for f in frames])
        _lib.check(self.lib.slm_bind_frames(self.h, first, len(frames), arr, self.stream), "slm_bind_frames")
        for i, f in enumerate(frames):
            self._frames[first + i] = f

    def run(self, n_frames: int):
        _lib.check(self.lib.slm_run(self.h, n_frames, self.stream), "slm_run")

    def beta(self, slot: int, out: torch.Tensor | None = None) -> torch.Tensor:
        J = self._frames[slot].J
        if out is None:
            out = torch.empty((J, 7), dtype=torch.float64, device=self.device)
        _lib.check(self.lib.slm_get_beta(self.h, slot, out.data_ptr(), self.stream), "slm_get_beta")
        return out

    def apply_update(self, slot: int, beta: torch.Tensor):
        """``Surfels.update`` in place on the slot's frame tensors."""
        f = self._frames[slot]
        fn = self.lib.slm_apply_update_f64 if f.sf_points.dtype == torch.float64 else self.lib.slm_apply_update
        _lib.check(fn(f.N, f.J, int(f.sf_knn_idx.shape[1]),
                                             f.sf_points.data_ptr(), f.sf_norms.data_ptr(),
                                             f.sf_knn_idx.data_ptr(), f.sf_knn_w.data_ptr(),
                                             f.ed_points.data_ptr(), f.ed_norms.data_ptr(),
                                             beta.data_ptr(), self.stream), "slm_apply_update")

    def records(self, slot: int):
        n = int(self.cfg.num_iterations)
        arr = (SlmIterRecord * max(n, 1))()
        _lib.check(self.lib.slm_get_records(self.h, slot, arr, n, self.stream), "slm_get_records")
        return [dict(loss=r.loss, u=r.u, accepted=bool(r.accepted), status=int(r.status),
                     M_grad=int(r.M_grad), M_loss=int(r.M_loss)) for r in arr[:n]]

    def plan_info(self, slot: int = 0):
        out = (C.c_double * _lib.PLAN_INFO_DOUBLES)()
        _lib.check(self.lib.slm_get_plan_info(self.h, slot, out, _lib.PLAN_INFO_DOUBLES), "slm_get_plan_info")
        keys = ("solver", "fronts", "levels", "factor_flops", "factor_bytes", "tuples", "runs", "pairs",
                "merged_records", "positions", "factor_flops_unpadded", "solver_tasks", "pivot_tiles", "pure_fill_tiles")
        d = {k: out[i] for i, k in enumerate(keys)}
        d["solver"] = "nested-dissection multifrontal" if d["solver"] == 0 else "band"
        return d

    def profile(self, on: bool):
        _lib.check(self.lib.slm_profile_enable(self.h, int(on)), "slm_profile_enable")

    def profile_read(self):
        n = len(_lib.PHASES)
        ms = (C.c_double * n)()
        cnt = (C.c_int64 * n)()
        _lib.check(self.lib.slm_profile_read(self.h, ms, cnt), "slm_profile_read")
        return {p: dict(ms=ms[i], count=int(cnt[i])) for i, p in enumerate(_lib.PHASES)}
