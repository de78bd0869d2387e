"""Diagnostic: end-to-end ms/frame of the stage mirrors in the reference driver's order (tests/driver_harness.py) at the SuPer image size:
depth_preprocessing -> LM (10 iterations) -> Surfels.update -> fuseInputData -> swap, per stage."""
import sys, os, time
# Host thread pools: on the GPU box (256 logical CPUs, cgroup quota of 16 CPUs per 100 ms) the default OpenMP /
# BLAS pools of torch and numpy (256 spinning threads after every small CPU op) exhaust the quota and the whole
# process is CFS-throttled for ~25 ms every 100 ms -- every third frame took 45 instead of 19 ms.  The library
# itself starts no threads; cap the pools like any latency-sensitive host program would.  DRIVER_THREADS=0 keeps
# the defaults (to reproduce the stalls; nr_throttled of /sys/fs/cgroup/cpu.stat is printed at the end).
_thr = os.environ.get("DRIVER_THREADS", "4")
if _thr != "0":
    for _k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
        os.environ.setdefault(_k, _thr)
from types import SimpleNamespace
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-super_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
from super_amd import synth, fusion, nodes
import driver_harness as drv
from super_amd import data_loader

H, W = 480, 640
derived = "--gf" not in sys.argv
step = 11
K = synth.intrinsics()
inv_K = np.linalg.pinv(K)
vv, uu = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
color = np.random.default_rng(2).uniform(0, 1, (3, H, W)).astype(np.float32)
opt = SimpleNamespace(height=H, width=W, data="superv1", load_valid_mask=False, depth_model="monodepth2",
                      dilate_invalid_kernel=0, normal_model="naive", phase="test", method="super", load_depth=True,
                      deform_udpate_method="super_edg", mesh_step_size=step, use_derived_gradient=derived,
                      sf_point_plane=True, mesh_arap=True, mesh_rot=True, mesh_face=False, sf_point_plane_weight=1.0,
                      mesh_arap_weight=10.0, mesh_rot_weight=1.0, mesh_face_weight=1.0, num_optimize_iterations=10,
                      optimizer="Adam", learning_rate=2e-4, num_neighbors=4, num_ED_neighbors=4, th_dist=0.02,
                      th_cosine_ang=0.4, th_time_steps=30, disable_merging_new_surfels=False,
                      disable_merging_exist_surfels=False, disable_adding_new_surfels=False,
                      disable_removing_unstable_surfels=False)
if os.environ.get("DRIVER_PREPARE_AHEAD") == "0":      # A/B: the whole prepare inside LM(), as in rounds 1-3
    opt.slm_prepare_ahead = False
if os.environ.get("DRIVER_SOLVER_PATH"):
    opt.slm_solver_path = int(os.environ["DRIVER_SOLVER_PATH"])
model = drv.FrameLoop(opt)
stages = {}


def cpu_stat():
    try:
        return {k: int(v) for k, v in (l.split() for l in open("/sys/fs/cgroup/cpu.stat"))}
    except OSError:
        return {}


stat0 = cpu_stat()


def timed(name, fn):
    def wrapper(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize()
        stages.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
        return r
    return wrapper


drv.depth_preprocessing = timed("depth_preprocessing", data_loader.depth_preprocessing)
drv.SurfelModel.update = timed("update", nodes.update)
drv.SurfelModel.fuseInputData = timed("fuseInputData", fusion.fuseInputData)
drv.SurfelModel.prepareStableIndexNSwapAllModel = timed("swap", fusion.prepareStableIndexNSwapAllModel)
if derived:
    model.lm.LM = timed("LM", model.lm.LM)
    if "--detail" in sys.argv:
        from super_amd import LM as lm_mod
        _bf = lm_mod.BoundFrame
        lm_mod.BoundFrame = timed("LM.BoundFrame (conversions)", _bf)
        _lib0 = model.lm.lib
        class _Wrap:
            def __init__(self, lib):
                self._lib = lib
            def __getattr__(self, name):
                fn = getattr(self._lib, name)
                if name in ("slm_bind_frame", "slm_run", "slm_get_beta", "slm_get_records"):
                    return timed("C " + name, fn)
                return fn
        model.lm.lib = _Wrap(_lib0)
else:
    model.graph_fit.forward = timed("GraphFit", model.graph_fit.forward)
    model.graph_fit.__class__.__call__ = lambda self, *a, **k: self.forward(*a, **k)
frames = int(os.environ.get("DRIVER_FRAMES", "12"))
tot = []
events = []
accept_log = []
if "--nogc" in sys.argv:
    import gc
    gc.disable()
for k in range(frames):
    depth = (0.2 * synth._surface(uu, vv, H, W, 0.3 + 0.3 * np.sin(0.05 * k))).astype(np.float32)
    depth[:4] = 0.0
    depth[:, :4] = 0.0
    inputs = {("depth", 0): torch.from_numpy(depth)[None, None].cuda(), ("disp", 0): torch.zeros(1, 1, H, W).cuda(),
              "inv_K": torch.from_numpy(inv_K)[None].cuda(), "K": torch.from_numpy(K)[None].cuda(),
              ("color", 0): torch.from_numpy(color)[None].cuda(), "divterm": torch.tensor(1.0 / (2 * 0.6 * 0.6)),
              "filename": ["%06d" % k], "time": k, "ID": torch.tensor([k])}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model(SimpleNamespace(), inputs)
    torch.cuda.synchronize()
    tot.append((time.perf_counter() - t0) * 1e3)
    if derived and model.lm.last_records:
        accept_log.append("".join("T" if r["accepted"] else ("F" if r["status"] == 0 else "x") for r in model.lm.last_records[0]))
    import ctypes
    from super_amd import _lib
    cnt = (ctypes.c_int64 * 4)()
    _lib.load().slm_debug_counters(cnt)
    events.append(tuple(cnt))
print("surfels", int(model.sf.points.shape[0]), "nodes", model.sf.ED_nodes.num, "path", "LM" if derived else "GraphFit",
      "| torch reserved MB", torch.cuda.memory_reserved() >> 20, "| device used MB",
      (torch.cuda.mem_get_info()[1] - torch.cuda.mem_get_info()[0]) >> 20)
print("ms/frame (frames 4..):", round(float(np.mean(tot[4:])), 2), "all:", [round(t, 1) for t in tot[:40]])
tt = np.array(tot[4:])
print(f"frame time over {len(tt)} frames: median {np.median(tt):.2f} ms, p90 {np.percentile(tt, 90):.2f}, p99 {np.percentile(tt, 99):.2f}, "
      f"max {tt.max():.2f}; p99 / median = {np.percentile(tt, 99) / np.median(tt):.3f}")
ev = np.array(events)
d = np.diff(ev, axis=0)
slow = [i + 1 for i in range(len(d)) if tot[i + 1] > 1.3 * np.median(tot)]
stat1 = cpu_stat()
if stat1:
    print("host: torch threads", torch.get_num_threads(), "| cgroup cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip(),
          "| CFS periods throttled during the run:", stat1.get("nr_throttled", 0) - stat0.get("nr_throttled", 0),
          "| throttled ms:", (stat1.get("throttled_usec", 0) - stat0.get("throttled_usec", 0)) // 1000)
print("slow frames:", [(i, round(tot[i], 1), "reallocs", int(d[i - 1][0]), "MB", int(d[i - 1][1]) >> 20, "plan builds", int(d[i - 1][2])) for i in slow[:25]])
print("totals: reallocs", ev[-1][0], "plan builds", ev[-1][2], "plan reuses", ev[-1][3])
if accept_log:
    n_it = sum(len(a) for a in accept_log)
    n_rej = sum(a.count("F") for a in accept_log)
    n_rej_followed = sum(sum(1 for i, c in enumerate(a[:-1]) if c == "F") for a in accept_log)
    print(f"LM accept patterns: {n_rej} of {n_it} iterations rejected ({100.0 * n_rej / max(n_it, 1):.1f} %), {n_rej_followed} of them "
          f"followed by another iteration of the same frame (what a speculative second solve could save); first frames: {accept_log[:12]}")
for k, v in stages.items():
    print(f"  {k:22s} {np.mean(v[3:]):7.2f} ms   per frame: {[round(x, 1) for x in v]}")
