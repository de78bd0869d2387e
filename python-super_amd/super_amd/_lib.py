"""ctypes binding of libsuper_lm.so (the C ABI declared in include/super_lm.h).

There is no fallback: if the shared library is missing or a call fails, this module
raises -- the product path never routes through a CPU or PyTorch implementation.
"""
from __future__ import annotations

import ctypes as C
import os

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG_ROOT, "lib", os.environ.get("SLM_LIB", "libsuper_lm.so"))

SLM_OK = 0
SLM_ERR_INVALID, SLM_ERR_HIP, SLM_ERR_NO_DEVICE, SLM_ERR_UNBOUND, SLM_ERR_UNSUPPORTED = 1, 2, 3, 4, 5   # include/super_lm.h:48-53
SLM_ITER_OK, SLM_ITER_SOLVER_FAILED, SLM_ITER_NOT_RUN, SLM_ITER_SOLVER_TIMEOUT = 0, 1, 2, 3
SLM_ABI_VERSION = 3          # SLM_ABI_VERSION of include/super_lm.h this binding was written against
PLAN_INFO_DOUBLES = 14
SLM_X_PAIR_BLOCKS, SLM_X_DELTA, SLM_X_DATA_LOSS = 0, 1, 2
PHASES = ["zero", "data_grad", "reg_grad", "solve", "data_loss", "accept"]

GF_NTERMS = 10     # SLM_GF_NTERMS of include/super_lm.h

EXPORTS = [
    "slm_create", "slm_destroy", "slm_last_error", "slm_device_count", "slm_bind_frame", "slm_bind_frames",
    "slm_run", "slm_profile_enable", "slm_profile_read", "slm_get_plan_info", "slm_get_beta", "slm_set_beta", "slm_get_records", "slm_assemble", "slm_loss",
    "slm_solve", "slm_solve_dense", "slm_data_residuals", "slm_apply_update", "slm_knn",
    "slm_knn_weights", "slm_gf_create", "slm_gf_destroy", "slm_gf_bind_frame", "slm_gf_run",
    "slm_gf_bind_semantic", "slm_gf_bind_flow", "slm_gf_get_edge_points", "slm_gf_set_shard", "slm_gf_eval_morph",
    "slm_gf_eval_losses", "slm_gf_step", "slm_gf_get_partial", "slm_gf_set_partial",
    "slm_depth_create", "slm_depth_destroy", "slm_depth_preprocess",
    "slm_graph_init", "slm_fuse_create", "slm_fuse_destroy", "slm_fuse_input_data", "slm_fuse_swap_stable",
    "slm_fuse_bind_semantic", "slm_knn_f64", "slm_knn_weights_f64", "slm_graph_init_semantic", "slm_debug_counters", "slm_debug_last_solver_form", "slm_debug_last_dag_mode",
    "slm_set_shard", "slm_lm_grad_local", "slm_lm_solve", "slm_lm_loss_local", "slm_lm_accept",
    "slm_lm_exchange_size", "slm_lm_exchange_get", "slm_lm_exchange_set", "slm_lm_exchange_ptr",
    "slm_gf_get_deform", "slm_gf_loss_grad", "slm_apply_update_gf",
    "slm_apply_update_f64", "slm_apply_update_gf_f64", "slm_debug_read", "slm_debug_dag_trace",
    "slm_abi_version", "slm_abi_check", "slm_debug_dag_timeout", "slm_debug_dag_abort", "slm_prepare_model", "slm_discard_prepared", "slm_debug_read_plan",
]


class SlmConfig(C.Structure):
    _fields_ = [("num_iterations", C.c_int32), ("phase_test", C.c_int32), ("use_data", C.c_int32),
                ("use_arap", C.c_int32), ("use_rot", C.c_int32), ("max_frames", C.c_int32),
                ("data_path", C.c_int32), ("solver_path", C.c_int32),
                ("w_data", C.c_double), ("w_arap", C.c_double), ("w_rot", C.c_double),
                ("u0", C.c_double), ("v", C.c_double), ("minimal_loss0", C.c_double)]


class SlmFrame(C.Structure):
    _fields_ = [("N", C.c_int32), ("J", C.c_int32), ("T", C.c_int32), ("H", C.c_int32),
                ("W", C.c_int32), ("K", C.c_int32), ("K_ED", C.c_int32),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("sf_points", C.c_void_p), ("sf_knn_idx", C.c_void_p), ("sf_knn_w", C.c_void_p),
                ("ed_points", C.c_void_p), ("ed_knn_idx", C.c_void_p), ("tgt_points", C.c_void_p),
                ("tgt_norms", C.c_void_p), ("index_map", C.c_void_p), ("tgt_valid", C.c_void_p),
                ("state_f64", C.c_int32), ("pad", C.c_int32)]


class SlmGfConfig(C.Structure):
    _fields_ = [("num_iterations", C.c_int32), ("optimizer", C.c_int32), ("use_data", C.c_int32),
                ("use_arap", C.c_int32), ("use_rot", C.c_int32), ("use_face", C.c_int32),
                ("max_frames", C.c_int32), ("seg_mode", C.c_int32), ("use_bn_morph", C.c_int32),
                ("corr_mode", C.c_int32),
                ("w_data", C.c_double), ("w_arap", C.c_double), ("w_rot", C.c_double),
                ("w_face", C.c_double), ("lr", C.c_double), ("w_bn_morph", C.c_double),
                ("pp_max", C.c_double), ("w_corr", C.c_double)]


class SlmGfFrame(C.Structure):
    _fields_ = [("base", SlmFrame), ("sf_stable", C.c_void_p), ("ed_knn_w", C.c_void_p),
                ("ed_triangles", C.c_void_p), ("ed_triangle_areas", C.c_void_p),
                ("n_triangles", C.c_int32), ("pad", C.c_int32)]


class SlmGfSemantic(C.Structure):
    _fields_ = [("num_classes", C.c_int32), ("pad", C.c_int32), ("sf_seg", C.c_void_p),
                ("sf_seg_conf", C.c_void_p), ("tgt_seg_conf", C.c_void_p),
                ("img_seg_conf", C.c_void_p), ("img_seg", C.c_void_p)]


class SlmDepthConfig(C.Structure):
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("data_mode", C.c_int32), ("raft_stereo", C.c_int32),
                ("dilate_invalid_kernel", C.c_int32), ("load_depth", C.c_int32), ("normal_model", C.c_int32),
                ("num_classes", C.c_int32), ("n_del_classes", C.c_int32), ("del_classes", C.c_int32 * 3),
                ("depth_width_range", C.c_float * 2), ("inv_K", C.c_float * 9),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("divterm", C.c_double), ("use_ssim_conf", C.c_int32), ("stereo_P", C.c_float * 12)]


class SlmDepthInputs(C.Structure):
    _fields_ = [("depth", C.c_void_p), ("color", C.c_void_p), ("valid_mask", C.c_void_p),
                ("seg", C.c_void_p), ("seg_conf", C.c_void_p)]


class SlmDepthOutputs(C.Structure):
    _fields_ = [("points", C.c_void_p), ("norms", C.c_void_p), ("colors", C.c_void_p),
                ("radii", C.c_void_p), ("confs", C.c_void_p), ("index_map", C.c_void_p),
                ("valid", C.c_void_p), ("seg", C.c_void_p), ("seg_conf", C.c_void_p),
                ("dist2edge", C.c_void_p), ("inval", C.c_void_p), ("disp_conf", C.c_void_p)]


class SlmFuseConfig(C.Structure):
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("merge_new", C.c_int32), ("merge_exist", C.c_int32),
                ("add_new", C.c_int32), ("remove_unstable", C.c_int32), ("phase_test", C.c_int32),
                ("th_time_steps", C.c_int32), ("th_dist", C.c_double), ("th_cosine_ang", C.c_double),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float)]


class SlmSurfelModel(C.Structure):
    _fields_ = [("n", C.c_int32), ("cap", C.c_int32), ("points", C.c_void_p), ("norms", C.c_void_p),
                ("colors", C.c_void_p), ("radii", C.c_void_p), ("confs", C.c_void_p),
                ("time_stamp", C.c_void_p), ("is_stable", C.c_void_p), ("knn_idx", C.c_void_p),
                ("knn_w", C.c_void_p), ("projdata", C.c_void_p), ("J", C.c_int32), ("K", C.c_int32),
                ("ed_points", C.c_void_p), ("ed_radii", C.c_void_p), ("merged_into", C.c_void_p)]


class SlmNewFrame(C.Structure):
    _fields_ = [("T", C.c_int32), ("time", C.c_int32), ("points", C.c_void_p), ("norms", C.c_void_p),
                ("colors", C.c_void_p), ("radii", C.c_void_p), ("confs", C.c_void_p), ("valid", C.c_void_p),
                ("index_map", C.c_void_p)]


SLM_MAX_CLASSES = 4   # include/super_lm.h


class SlmFuseSemantic(C.Structure):
    _fields_ = [("num_classes", C.c_int32), ("soft_weights", C.c_int32), ("hard_seg", C.c_int32),
                ("merge_same_class", C.c_int32), ("seg", C.c_void_p), ("seg_conf", C.c_void_p),
                ("dist2edge", C.c_void_p), ("ed_seg", C.c_void_p), ("ed_seg_conf", C.c_void_p),
                ("new_seg", C.c_void_p), ("new_seg_conf", C.c_void_p), ("new_dist2edge", C.c_void_p)]


class SlmGraphOutputs(C.Structure):
    _fields_ = [("cap_nodes", C.c_int32), ("pad", C.c_int32), ("points", C.c_void_p), ("norms", C.c_void_p),
                ("radii", C.c_void_p), ("edge_index", C.c_void_p), ("edges_lens", C.c_void_p),
                ("triangles", C.c_void_p), ("triangles_areas", C.c_void_p)]


class SlmIterRecord(C.Structure):
    _fields_ = [("loss", C.c_double), ("u", C.c_double), ("accepted", C.c_int32),
                ("status", C.c_int32), ("M_grad", C.c_int32), ("M_loss", C.c_int32)]


class SuperLMError(RuntimeError):
    pass


_lib = None


def load():
    """Load the shared library (once) and declare the signatures; raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    # torch ships its own libamdhip64.so.7; import it first so that this library binds to the
    # SAME HIP runtime instance (device pointers and streams are shared with torch).
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise SuperLMError(
            f"{LIB_PATH} not found: build it with `python python-super_amd/super_amd/build.py` "
            "(hipcc, gfx950). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, dbl = C.c_void_p, C.c_int32, C.c_double
    lib.slm_last_error.restype = C.c_char_p
    lib.slm_last_error.argtypes = []
    lib.slm_device_count.restype = C.c_int
    lib.slm_device_count.argtypes = []
    sig = {
        "slm_create": [C.POINTER(SlmConfig), C.POINTER(vp)],
        "slm_destroy": [vp],
        "slm_debug_counters": [C.POINTER(C.c_int64)],
        "slm_debug_read": [vp, i32, i32, vp, C.c_int64, C.POINTER(C.c_int64), vp],
        "slm_debug_dag_trace": [vp, i32, i32, vp],
        "slm_debug_last_solver_form": [vp],
        "slm_debug_last_dag_mode": [vp],
        "slm_bind_frames": [vp, i32, i32, C.POINTER(SlmFrame), vp],
        "slm_bind_frame": [vp, i32, C.POINTER(SlmFrame), vp],
        "slm_prepare_model": [vp, i32, C.POINTER(SlmFrame), vp],
        "slm_discard_prepared": [vp, i32],
        "slm_debug_read_plan": [vp, i32, i32, vp, C.c_int64, C.POINTER(C.c_int64), vp],
        "slm_run": [vp, i32, vp],
        "slm_profile_enable": [vp, i32],
        "slm_get_plan_info": [vp, i32, C.POINTER(C.c_double), i32],
        "slm_abi_check": [i32, i32, i32, i32, i32, i32],
        "slm_debug_dag_timeout": [C.c_int64],
        "slm_debug_dag_abort": [vp, i32, vp],
        "slm_profile_read": [vp, C.POINTER(C.c_double), C.POINTER(C.c_int64)],
        "slm_get_beta": [vp, i32, vp, vp],
        "slm_set_beta": [vp, i32, vp, vp],
        "slm_get_records": [vp, i32, C.POINTER(SlmIterRecord), i32, vp],
        "slm_assemble": [vp, i32, vp, vp, vp],
        "slm_loss": [vp, i32, vp, vp],
        "slm_solve": [vp, i32, dbl, vp, vp, vp],
        "slm_solve_dense": [i32, vp, vp, vp, vp, vp],
        "slm_data_residuals": [vp, i32, vp, vp, vp, vp],
        "slm_apply_update": [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp],
        "slm_knn": [i32, i32, i32, i32, vp, vp, vp, vp, vp],
        "slm_knn_weights": [i32, i32, i32, vp, vp, vp, vp, vp, vp],
        "slm_knn_f64": [i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp],
        "slm_knn_weights_f64": [i32, i32, i32, vp, vp, vp, i32, vp, vp, vp, vp, vp],
        "slm_gf_create": [C.POINTER(SlmGfConfig), C.POINTER(vp)],
        "slm_gf_destroy": [vp],
        "slm_gf_bind_frame": [vp, i32, C.POINTER(SlmGfFrame), vp],
        "slm_gf_run": [vp, i32, vp],
        "slm_gf_bind_semantic": [vp, i32, C.POINTER(SlmGfSemantic), C.POINTER(C.c_int32), vp],
        "slm_gf_bind_flow": [vp, i32, vp, vp],
        "slm_gf_get_edge_points": [vp, i32, i32, vp, i32, vp],
        "slm_gf_set_shard": [vp, i32, i32],
        "slm_gf_eval_morph": [vp, i32, vp],
        "slm_gf_eval_losses": [vp, i32, vp],
        "slm_gf_step": [vp, i32, vp],
        "slm_gf_get_partial": [vp, i32, vp, vp],
        "slm_gf_set_partial": [vp, i32, vp, vp],
        "slm_set_shard": [vp, i32, i32],
        "slm_lm_grad_local": [vp, i32, vp],
        "slm_lm_solve": [vp, i32, vp],
        "slm_lm_loss_local": [vp, i32, vp],
        "slm_lm_accept": [vp, i32, vp],
        "slm_lm_exchange_size": [vp, i32, i32, C.POINTER(C.c_int64)],
        "slm_lm_exchange_ptr": [vp, i32, i32, C.POINTER(vp), C.POINTER(C.c_int64)],
        "slm_lm_exchange_get": [vp, i32, i32, vp, vp],
        "slm_lm_exchange_set": [vp, i32, i32, vp, vp],
        "slm_graph_init": [i32, i32, i32, vp, vp, vp, vp, C.POINTER(SlmGraphOutputs), C.POINTER(C.c_int32), vp],
        "slm_graph_init_semantic": [i32, i32, i32, vp, vp, vp, vp, i32, vp, i32, C.POINTER(SlmGraphOutputs), vp, vp,
                                    C.POINTER(C.c_int32), vp],
        "slm_fuse_create": [i32, i32, i32, C.POINTER(vp)],
        "slm_fuse_destroy": [vp],
        "slm_fuse_bind_semantic": [vp, C.POINTER(SlmFuseSemantic)],
        "slm_fuse_input_data": [vp, C.POINTER(SlmFuseConfig), C.POINTER(SlmSurfelModel), C.POINTER(SlmNewFrame), vp],
        "slm_fuse_swap_stable": [vp, C.POINTER(SlmFuseConfig), C.POINTER(SlmSurfelModel), i32, vp, i32, vp, vp],
        "slm_depth_create": [i32, i32, C.POINTER(vp)],
        "slm_depth_destroy": [vp],
        "slm_depth_preprocess": [vp, C.POINTER(SlmDepthConfig), C.POINTER(SlmDepthInputs),
                                 C.POINTER(SlmDepthOutputs), C.POINTER(C.c_int32), vp],
        "slm_gf_get_deform": [vp, i32, vp, vp],
        "slm_gf_loss_grad": [vp, i32, vp, vp, vp, vp],
        "slm_apply_update_gf": [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp],
        "slm_apply_update_f64": [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp],
        "slm_apply_update_gf_f64": [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.restype = C.c_int
        fn.argtypes = args
    lib.slm_abi_version.restype = C.c_int
    lib.slm_abi_version.argtypes = []
    # ABI handshake: the struct mirrors above must be the library's own layouts
    rc = lib.slm_abi_check(SLM_ABI_VERSION, C.sizeof(SlmConfig), C.sizeof(SlmFrame), C.sizeof(SlmGfConfig),
                           C.sizeof(SlmGfFrame), C.sizeof(SlmIterRecord))
    if rc != SLM_OK:
        raise SuperLMError(lib.slm_last_error().decode(errors="replace"))
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != SLM_OK:
        msg = load().slm_last_error().decode(errors="replace")
        raise SuperLMError(f"{what} failed with status {rc}: {msg}")
