#!/usr/bin/env python
"""bench.py -- LM iterations/s and ms/frame of the embedded-deformation LM hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload C2] [--frames-per-gpu B]

One *step* = one pass of the hot path over one batch of synthetic input: for each of the
B independent frames resident on a GPU, bind the frame (``loss_term.prepare``), run
``num_optimize_iterations`` = 10 damped LM iterations on the device (Jacobian pass, multifrontal
Cholesky solve, loss pass, accept/reject) and apply ``Surfels.update``.  Inputs are in HBM
before the timed region starts.  N > 1: one process per GPU (torchrun), frames sharded
across ranks (weak scaling, B frames per GPU), end-of-frame RCCL all-gather of beta
(SURVEY.md section 8e); rank 0 prints ONE JSON line.

`value` = whole-job LM iterations per second.  `roofline` = the TIME-DOMINANT phase, the float64
multifrontal factor + substitutions (f64-MFMA bound; FLOPs from the plan, padded and unpadded),
timed with HIP events on its launch stream inside the timed region; `roofline_data` = the fused
data-term Jacobian pass (one launch per LM iteration, HBM bound); `whole_step_hbm_frac` = the
algorithmic bytes of the whole step against the HBM peak (the path is latency-bound: an exact
float64 solve is a chain of dependent tile factorisations); `latency_b1` = the same workload at
ONE frame per launch -- what a sequential tracker (run_super.py) sees; `cpu_baseline` = the NumPy
oracle (a port of the reference algorithm) timed on this box's host cores on a bounded sample;
`host.cpu_cores_busy_per_rank` = user + system CPU time of this rank's threads per second of the timed region (what a rank
costs of a node's CPU quota; `BENCH_THREAD_CPU=1` lists it per thread on stderr).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "python-super_amd"))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
F64_MFMA_PEAK_TFLOPS = 78.6  # SURVEY.md 8d (FP64 matrix, nominal gfx950)
NB = 64                      # tile edge of the banded solver (csrc/slm_common.h)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="C2", choices=["C1", "C2", "C4", "tiny"])
    ap.add_argument("--frames-per-gpu", type=int, default=8)
    ap.add_argument("--streams", type=int, default=1,
                    help="split the frames of a GPU over this many solver handles / HIP streams")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip HIP-event phase timing")
    ap.add_argument("--no-latency-b1", action="store_true",
                    help="skip the one-frame-per-launch leg (profiles/collect.sh: keeps the kernel statistics of a "
                         "profiled run to the launches of the headline configuration)")
    return ap.parse_args()


def workload_dims(name):
    from super_amd import synth
    if name == "tiny":
        return dict(N=3000, J=48, H=60, W=80, src_border=5, tgt_border=3)
    return dict(synth.WORKLOADS[name])


def lib_sha16():
    """sha256[:16] over the library's SOURCES (csrc/*.hip, csrc/*.h, include/*.h, sorted by name): what a committed
    profile is tagged with -- the built .so is not in the repository and need not be byte-reproducible."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "python-super_amd", "csrc", "*.hip")) +
                   glob.glob(os.path.join(ROOT, "python-super_amd", "csrc", "*.h")) +
                   glob.glob(os.path.join(ROOT, "include", "*.h")))
    if not files:
        return None
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel, workload, frames_per_gpu):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC summary
    (profiles/*pmc_traffic.json, made by profiles/make_traffic.py from separate FETCH_SIZE /
    WRITE_SIZE passes with the gfx950 corrections).  Returns (bytes or None, provenance): a summary
    is only used when it was collected for THIS workload / batch; the provenance names the file and
    says whether the library sources it profiled are identical to the ones in the tree now (`stale` otherwise:
    the figure then describes an older build of the kernel and is reported as null)."""
    import glob
    best, src = None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("workload") == workload and d.get("frames_per_gpu") == frames_per_gpu \
                and kernel in d.get("kernels", {}):
            best = d["kernels"][kernel]["traffic_bytes_per_launch"]
            src = {"file": os.path.relpath(path, ROOT), "lib_sha16": d.get("lib_sha16"),
                   "stale": d.get("lib_sha16") != lib_sha16()}
    if src is not None and src["stale"]:
        best = None
    return best, src


def pmc_phase_traffic(phase, workload, frames_per_gpu):
    """HBM bytes per LM iteration of one phase (all its launches) from the newest committed PMC summary of THIS
    workload / batch (profiles/make_traffic.py `phases`); (None, provenance) when the summary is of other sources."""
    import glob
    best, src = None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc_traffic.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if d.get("workload") == workload and d.get("frames_per_gpu") == frames_per_gpu and phase in d.get("phases", {}):
            best = d["phases"][phase]["traffic_bytes_per_iteration"]
            src = {"file": os.path.relpath(path, ROOT), "lib_sha16": d.get("lib_sha16"),
                   "stale": d.get("lib_sha16") != lib_sha16(), "kernels": d["phases"][phase].get("kernels")}
    if src is not None and src["stale"]:
        best = None
    return best, src


def rows_traffic(key):
    """HBM bytes of one slm_gf_run at C2 (`key` = "b1" / "b8") from the newest committed PMC summary of tools/profile_rows.py
    (profiles/*rows_pmc_traffic.json, profiles/make_rows_traffic.py); (None, provenance) when it describes other sources."""
    import glob
    best, src = None, None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*rows_pmc_traffic.json"))):
        try:
            d = json.load(open(path))
        except Exception:
            continue
        if key in d.get("graphfit_c2", {}):
            best = d["graphfit_c2"][key]["traffic_bytes_per_run"]
            src = {"file": os.path.relpath(path, ROOT), "lib_sha16": d.get("lib_sha16"), "stale": d.get("lib_sha16") != lib_sha16()}
    if src is not None and src["stale"]:
        best = None
    return best, src


def cpu_quota():
    """CPUs the container may use per scheduling period (cgroup v2 cpu.max / v1 cfs quota), or None = unlimited:
    os.cpu_count() reports the machine, not the quota, and a host pool sized by it gets throttled."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / per
    except Exception:
        return None


def thread_cpu():
    """{tid: (user + system CPU seconds, name)} of every thread of this process (/proc/self/task)."""
    out = {}
    tck = os.sysconf("SC_CLK_TCK")
    for t in os.listdir("/proc/self/task"):
        try:
            f = open(f"/proc/self/task/{t}/stat").read()
            name = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            out[int(t)] = ((int(rest[11]) + int(rest[12])) / tck, name)
        except (OSError, ValueError):
            pass
    return out


def host_info():
    model = None
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        affinity = len(os.sched_getaffinity(0))
    except Exception:
        affinity = None
    return {"cpu_model": model, "logical_cpus": os.cpu_count(), "cpu_affinity": affinity, "cpu_quota": cpu_quota()}


def latency_b1(dims, device, iters, steps=6):
    """The same workload at ONE frame per launch (what `run_super.py` hits: one sequential frame at a
    time): bind + 10 LM iterations + Surfels.update, timed like the main loop, with the phase split."""
    import torch
    from super_amd import synth
    from super_amd.engine import DeviceFrame, Engine
    sc = synth.make_scene(seed=0, **dims)
    pristine, work = DeviceFrame.from_scene(sc, device), DeviceFrame.from_scene(sc, device)
    eng = Engine(device, max_frames=1, num_iterations=iters)
    beta = torch.empty((sc.J, 7), dtype=torch.float64, device=device)

    def step():
        work.sf_points.copy_(pristine.sf_points)
        work.sf_norms.copy_(pristine.sf_norms)
        work.ed_points.copy_(pristine.ed_points)
        work.ed_norms.copy_(pristine.ed_norms)
        eng.bind(0, work)
        eng.run(1)
        eng.beta(0, beta)
        eng.apply_update(0, beta)

    for _ in range(2):
        step()
    torch.cuda.synchronize(device)
    eng.profile(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(device)
    dt = (time.perf_counter() - t0) / steps
    prof = eng.profile_read()
    eng.profile(False)
    info = eng.plan_info(0)
    phases = {k: v["ms"] / max(v["count"], 1) for k, v in prof.items()}
    solve_s = phases["solve"] * 1e-3
    out = {"value": iters / dt, "unit": "LM it/s", "ms_per_frame": 1e3 * dt, "frames_per_launch": 1,
           "phase_ms_per_iteration": phases,
           "solver_form": "task graph (one persistent launch)" if info["solver_tasks"] > 0 else "per-level launches",
           "solver_tflops": info["factor_flops_unpadded"] / solve_s / 1e12 if solve_s > 0 else 0.0,
           "solver_frac_of_f64_mfma_peak": info["factor_flops_unpadded"] / solve_s / 1e12 / F64_MFMA_PEAK_TFLOPS if solve_s > 0 else 0.0,
           "solver_tflops_padded": info["factor_flops"] / solve_s / 1e12 if solve_s > 0 else 0.0,
           "solver_frac_of_f64_mfma_peak_padded": info["factor_flops"] / solve_s / 1e12 / F64_MFMA_PEAK_TFLOPS if solve_s > 0 else 0.0,
           "sample": f"{steps} steps of one {sc.N}-surfel / {sc.J}-node frame"}
    eng.close()
    return out


def batch_depth(scenes, device, iters, depths=(16, 32), steps=3):
    """The headline's step (bind + 10 LM iterations + Surfels.update per frame) at DEEPER batches than the headline's 8 frames per
    launch -- BASELINE `configs[3]`'s 64 frames on ONE GPU would be two launches of 32.  The pivot chains of the solve are
    latency-bound and shared by every frame of a launch, so the rate grows with the depth; reported beside the headline, never as it.
    The frames are the bench's own scenes, repeated over the slots (every slot owns its state arrays)."""
    import torch
    from super_amd.engine import DeviceFrame, Engine
    fields = ("sf_points", "sf_norms", "ed_points", "ed_norms")
    out = {}
    for depth in depths:
        pristine = [DeviceFrame.from_scene(scenes[i % len(scenes)], device) for i in range(depth)]
        work = [DeviceFrame.from_scene(scenes[i % len(scenes)], device) for i in range(depth)]
        eng = Engine(device, max_frames=depth, num_iterations=iters)
        betas = [torch.empty((scenes[0].J, 7), dtype=torch.float64, device=device) for _ in range(depth)]

        def step():
            for p, w in zip(pristine, work):
                for nm in fields:
                    getattr(w, nm).copy_(getattr(p, nm))
            eng.bind_batch(work)
            eng.run(depth)
            for i in range(depth):
                eng.beta(i, betas[i])
                eng.apply_update(i, betas[i])

        for _ in range(2):
            step()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize(device)
        dt = (time.perf_counter() - t0) / steps
        ok = all(r["status"] == 0 for i in range(depth) for r in eng.records(i))
        out[str(depth)] = {"value": depth * iters / dt, "unit": "LM it/s", "ms_per_step": 1e3 * dt, "ms_per_frame": 1e3 * dt / depth,
                           "frames_per_launch": depth, "all_iterations_ok": ok}
        eng.close()
        del pristine, work, betas
    out["sample"] = f"{steps} steps per depth; the {len(scenes)} frames of the headline repeated over the slots"
    return out


def latency_b1_sequence(device, frames=48):
    """`latency_b1_sequence`: the one-frame-per-launch path as a TRACKER sees it -- not the always-warm re-bind of one
    frame that `latency_b1` times, but a moving surface at the SuPer image size (480 x 640, about 300 k surfels / 2.5 k
    grid-mesh nodes) through the stage mirrors in the reference driver's order (tests/driver_harness.py: depth
    preprocessing -> LM -> update -> fusion -> swap; reference super/super.py:23-73): surfels come and go, the coupling
    graph changes, some frames need a new symbolic analysis.  Host-timed per stage around stream syncs.  Run twice: with the
    model-side prepare started ahead of the next frame (`slm_prepare_model`, after the swap) and with the whole prepare
    inside LM() as in rounds 1-3."""
    import ctypes as C
    import numpy as np
    import torch
    from types import SimpleNamespace
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import driver_harness as drv
    from super_amd import _lib, synth
    H, W = 480, 640
    K = synth.intrinsics()
    inv_K = np.linalg.pinv(K)
    vv, uu = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    color = np.random.default_rng(2).uniform(0, 1, (3, H, W)).astype(np.float32)
    depths = []
    for k in range(frames):
        d = (0.2 * synth._surface(uu, vv, H, W, 0.3 + 0.3 * np.sin(0.05 * k))).astype(np.float32)
        d[:4] = 0.0
        d[:, :4] = 0.0
        depths.append(torch.from_numpy(d)[None, None])
    lib = _lib.load()
    cnt = (C.c_int64 * 4)()
    out = {}
    for label, ahead in (("prepare_ahead", True), ("prepare_inside_lm", False)):
        opt = SimpleNamespace(height=H, width=W, data="superv1", load_valid_mask=False, depth_model="monodepth2",
                              dilate_invalid_kernel=0, normal_model="naive", phase="test", method="super", load_depth=True,
                              deform_udpate_method="super_edg", mesh_step_size=11, use_derived_gradient=True,
                              sf_point_plane=True, mesh_arap=True, mesh_rot=True, mesh_face=False, sf_point_plane_weight=1.0,
                              mesh_arap_weight=10.0, mesh_rot_weight=1.0, mesh_face_weight=1.0, num_optimize_iterations=10,
                              num_neighbors=4, num_ED_neighbors=4, th_dist=0.02, th_cosine_ang=0.4, th_time_steps=30,
                              disable_merging_new_surfels=False, disable_merging_exist_surfels=False,
                              disable_adding_new_surfels=False, disable_removing_unstable_surfels=False,
                              slm_prepare_ahead=ahead)
        loop = drv.FrameLoop(opt)
        lm_ms, tot_ms, builds, rej, its = [], [], [], 0, 0
        lm_call = loop.lm.LM

        def timed_lm(*a, **k):
            torch.cuda.synchronize(device)
            t0 = time.perf_counter()
            r = lm_call(*a, **k)
            torch.cuda.synchronize(device)
            lm_ms.append((time.perf_counter() - t0) * 1e3)
            return r
        loop.lm.LM = timed_lm
        for k in range(frames):
            inputs = {("depth", 0): depths[k].to(device), ("disp", 0): torch.zeros(1, 1, H, W, device=device),
                      "inv_K": torch.from_numpy(inv_K)[None].to(device), "K": torch.from_numpy(K)[None].to(device),
                      ("color", 0): torch.from_numpy(color)[None].to(device), "divterm": torch.tensor(1.0 / (2 * 0.6 * 0.6)),
                      "filename": ["%06d" % k], "time": k, "ID": torch.tensor([k])}
            torch.cuda.synchronize(device)
            lib.slm_debug_counters(cnt)
            b0 = cnt[2]
            t0 = time.perf_counter()
            loop(SimpleNamespace(), inputs)
            torch.cuda.synchronize(device)
            tot_ms.append((time.perf_counter() - t0) * 1e3)
            if k + 1 < frames:                          # the next frame's acquisition: a preparation queued after the swap runs under it
                time.sleep(0.004)
            lib.slm_debug_counters(cnt)
            builds.append(int(cnt[2] - b0))
            if loop.lm.last_records and k > 0:
                recs = loop.lm.last_records[0]
                its += len(recs)
                rej += sum(1 for r in recs if r["status"] == 0 and not r["accepted"])
        skip = 4                                         # first frames: allocations, the first plan
        lm = np.array(lm_ms[skip - 1:])                  # (no LM on frame 0)
        tt = np.array(tot_ms[skip:])
        q = lambda a, p: float(np.percentile(a, p))
        out[label] = {"lm_stage_ms": {"median": q(lm, 50), "p99": q(lm, 99), "mean": float(lm.mean())},
                      "frame_ms": {"median": q(tt, 50), "p99": q(tt, 99), "mean": float(tt.mean())},
                      "symbolic_analyses": int(sum(builds[skip:])),
                      "frames_with_a_symbolic_analysis": float(np.mean([b > 0 for b in builds[skip:]])),
                      "rejected_iteration_share": rej / max(its, 1)}
        surf, nodes = int(loop.sf.points.shape[0]), int(loop.sf.ED_nodes.num)
        del loop
    out["sample"] = (f"{frames} frames of a moving synthetic surface, {H}x{W}, {surf} surfels / {nodes} nodes at the end, 10 LM "
                     "iterations per frame, one frame per launch (task-graph solver); the first 4 frames are not counted; 4 ms of host "
                     "idle time between frames stand for the acquisition of the next frame (a depth network takes far longer)")
    out["note"] = ("`prepare_ahead`: slm_prepare_model right after the swap -- the sort, the size read-backs and the symbolic analyses "
                   "run on the library's worker thread between frames, LM() only binds the target; `prepare_inside_lm`: rounds 1-3")
    return out


def bind_timing(dims, device, B, reps=6):
    """The per-frame prepare (slm_bind_frame / slm_bind_frames = loss_term.prepare, reference super/loss.py:212-220,
    408-426) on its own, host-timed around a stream sync: `warm` = the frame's coupled-pair list is the one the slot's
    symbolic plan was built for (what the bench's own steps see: same frames every step); `cold` = every bind needs a new
    symbolic analysis on the host (the node-KNN table alternates between two orderings of the same neighbour sets: another
    hash, the same graph, so the work is that of a genuinely new plan with the buffers already allocated)."""
    import ctypes as C
    import torch
    from super_amd import synth
    from super_amd.engine import DeviceFrame, Engine
    out = {}
    cnt = (C.c_int64 * 4)()
    for label, nb in (("b1", 1), ("batch", B)):
        scs = [synth.make_scene(seed=i, **dims) for i in range(nb)]
        fa = [DeviceFrame.from_scene(sc, device) for sc in scs]
        fb = [DeviceFrame.from_scene(sc, device) for sc in scs]
        for f in fb:
            f.ed_knn_idx = torch.roll(f.ed_knn_idx, 1, dims=1).contiguous()
        eng = Engine(device, max_frames=nb, num_iterations=1)
        eng.bind_batch(fa)
        eng.bind_batch(fb)                                  # (both variants have been seen: allocations done)
        for mode in ("warm", "cold"):
            eng.bind_batch(fa)
            torch.cuda.synchronize(device)
            eng.lib.slm_debug_counters(cnt)
            builds0 = cnt[2]
            t0 = time.perf_counter()
            for r in range(reps):
                eng.bind_batch(fb if (mode == "cold" and r % 2 == 0) else fa)
            torch.cuda.synchronize(device)
            dt = (time.perf_counter() - t0) / reps
            eng.lib.slm_debug_counters(cnt)
            out[f"{mode}_ms_per_frame_{label}"] = 1e3 * dt / nb
            out[f"{mode}_symbolic_analyses_per_bind_{label}"] = (cnt[2] - builds0) / (reps * nb)
        eng.close()
    out["note"] = ("host-timed incl. the stream sync; `batch` = slm_bind_frames over the bench's frames per GPU (workers inside "
                   "the library), `b1` = one slm_bind_frame; the bench's timed steps contain the warm batch bind")
    return out


def cpu_baseline(dims, seed):
    """Oracle (NumPy/SciPy float64 port of the reference LM path) on the host cores:
    THREE LM iterations (Jacobian pass + dense Cholesky solve + loss pass each) of one frame,
    about 10 s of CPU work on the GPU box."""
    import numpy as np
    from oracle import lm_oracle as orc
    from super_amd import synth
    sc = synth.make_scene(seed=seed, **dims)
    fr = orc.Frame.from_scene(sc)
    n_it = 3
    opt = orc.default_opt(num_optimize_iterations=n_it)
    # The BLAS pool defaults to one thread per LOGICAL CPU of the machine (256 on the GPU box) whatever the cgroup
    # quota (16 CPUs there): the baseline is timed at the best thread count of a short probe -- a dense Cholesky of the
    # workload's own size, the part that dominates and the only multi-threaded one -- and `cores` is what was used.
    used, probe = None, {}
    try:
        from threadpoolctl import threadpool_info, threadpool_limits
        n_max = max([p.get("num_threads", 1) for p in threadpool_info()] + [1])
        import scipy.linalg as sla
        P = 7 * sc.J
        rng = np.random.default_rng(0)
        A = rng.standard_normal((P, 64))
        A = A @ A.T + P * np.eye(P)
        quota = cpu_quota()
        cands = sorted({c for c in (4, 8, 16, 32, 64, n_max, int(quota) if quota else n_max) if 1 <= c <= n_max})
        for n in cands:
            with threadpool_limits(limits=n):
                sla.cho_factor(A, lower=True, check_finite=False)           # warm-up at this thread count
                t0 = time.perf_counter()
                sla.cho_factor(A, lower=True, check_finite=False)
                probe[n] = time.perf_counter() - t0
        used = min(probe, key=probe.get)
        ctx = threadpool_limits(limits=used)
    except Exception:
        import contextlib
        ctx = contextlib.nullcontext()
        used = os.cpu_count() or 1
    trace = []
    with ctx:
        t0 = time.perf_counter()
        beta = orc.lm(fr, opt, trace=trace)
        dt = time.perf_counter() - t0
    return {"value": n_it / dt, "unit": "LM it/s", "cores": int(used), "kind": "port",
            "sample": f"{n_it} LM iterations (of 10) of one {sc.N}-surfel / {sc.J}-node frame, "
                      f"NumPy/SciPy float64 oracle incl. dense Cholesky, {dt:.1f} s; `cores` = BLAS pool threads used, the "
                      "best of `thread_probe_s` (seconds per dense Cholesky of this size at each thread count)",
            "thread_probe_s": {str(k): v for k, v in probe.items()},
            "seconds": dt}, beta, trace


def parity_check(dims, device, oracle_beta, trace, B):
    """`parity_<workload>`: the oracle's beta after the 3 LM iterations `cpu_baseline` has just computed (frame seed 0)
    against the HIP path on the same frame, same 3 iterations: at one frame per launch, and as slot 0 of a batch of the
    bench's own `frames_per_gpu` frames (seeds 0 .. B-1: the solver form and launch shapes of the timed loop).  The
    oracle is the checker here, never the thing timed; a mismatch fails the bench.  (All ten iterations of the first
    and the last slot of an 8-frame batch are compared in tests/test_gpu_bench_shape_parity.py.)"""
    import numpy as np
    from super_amd import synth
    from super_amd.engine import DeviceFrame, Engine
    n_it = len(trace)
    out = {"iterations": n_it, "tolerance": 1e-4, "reference": "oracle/lm_oracle.py (NumPy float64, dense Cholesky)"}
    worst = 0.0
    legs = [("one_frame_per_launch", 1)] + ([(f"batch_of_{B}", B)] if B > 1 else [])
    for label, nb in legs:
        eng = Engine(device, max_frames=nb, num_iterations=n_it)
        eng.bind_batch([DeviceFrame.from_scene(synth.make_scene(seed=i, **dims), device) for i in range(nb)])
        eng.run(nb)
        recs = eng.records(0)
        err = float(np.abs(eng.beta(0).cpu().numpy() - oracle_beta).max())
        loss_rel = max(abs(r["loss"] - t["loss"]) / max(abs(t["loss"]), 1e-300) for r, t in zip(recs, trace))
        out[label] = {"max_abs_beta_diff": err, "max_rel_loss_diff": loss_rel, "frames_per_launch": nb,
                      "match_counts_equal": [r["M_grad"] for r in recs] == [t["M_grad"] for t in trace],
                      "accept_flags_equal": [r["accepted"] for r in recs] == [bool(t["accepted"]) for t in trace],
                      "solver_form": {0: "per-level", 1: "task graph", 2: "hybrid"}.get(eng.lib.slm_debug_last_solver_form(eng.h))}
        worst = max(worst, err)
        eng.close()
    out["max_abs_beta_diff"] = worst
    out["ok"] = bool(worst < 1e-4 and all(out[k]["match_counts_equal"] and out[k]["max_rel_loss_diff"] < 1e-6
                                          for k, _ in legs))
    return out


def cpu_baseline_autograd(dims, seed):
    """The reference's DEFAULT optimiser (GraphFit: autograd + Adam, BASELINE configs[0]) as
    restated in oracle/graphfit_oracle.py (PyTorch CPU, float64): one full frame = 10 Adam
    iterations of point-plane + ARAP + Rot on one frame."""
    import torch
    from oracle import graphfit_oracle as gfo
    from super_amd import synth
    sc = synth.make_scene(seed=seed, **dims)
    pb = gfo.Problem(sc)
    opt = gfo.default_opt(optimizer="Adam")
    # PyTorch's intra-op pool gets slower past a few dozen threads on these 200k-element ops (GPU box,
    # 128 default threads: 0.76 s per Adam iteration; 16 threads: 0.13 s), so the baseline is timed
    # at the best of a short probe and `cores` reports the thread count actually used.
    n_thr = torch.get_num_threads()
    try:
        probe = gfo.default_opt(optimizer="Adam", num_optimize_iterations=1)
        best, used = None, n_thr
        for n in sorted({min(n_thr, c) for c in (8, 16, 32, 64, n_thr)}):
            torch.set_num_threads(n)
            gfo.graphfit(pb, probe)                       # warm-up at this thread count
            t0 = time.perf_counter()
            gfo.graphfit(pb, probe)
            dt1 = time.perf_counter() - t0
            if best is None or dt1 < best:
                best, used = dt1, n
        torch.set_num_threads(used)
        t0 = time.perf_counter()
        gfo.graphfit(pb, opt)
        dt = time.perf_counter() - t0
    finally:
        torch.set_num_threads(n_thr)
    return {"value": opt.num_optimize_iterations / dt, "unit": "Adam it/s", "cores": used,
            "kind": "port", "ms_per_frame": 1e3 * dt,
            "sample": f"10 Adam iterations (one frame) of the autograd path on one {sc.N}-surfel / "
                      f"{sc.J}-node frame, PyTorch-CPU float64, {dt:.1f} s"}


def _graphfit_frames(sc, device, semantic=False):
    """(sf, inputs, new_data) of one synthetic scene for the GraphFit mirror (the semantic fields when asked for)."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import torch_frame
    sf, inputs, new_data = torch_frame(sc, device)
    if semantic:
        t = lambda a: torch.from_numpy(a).to(device)
        sf.ED_nodes.triangles = t(sc.ed_triangles)
        sf.ED_nodes.triangles_areas = t(sc.ed_triangle_areas).double()
        sf.seg, sf.seg_conf = t(sc.sf_seg), t(sc.sf_seg_conf)
        new_data.seg_conf = t(sc.tgt_seg_conf)
        inputs[("seg_conf", 0)] = t(sc.img_seg_conf)[None]
        inputs[("seg", 0)] = t(sc.img_seg)[None, None]
    return sf, inputs, new_data


def _time_gf_run(gf, n_frames, device, reps=10):
    """ms per slm_gf_run(n_frames) from HIP events on the launch stream (the current torch stream): one event per run, the
    MEDIAN run (a single stalled run -- seen once: 80 ms inside ten runs of 2 ms on a fresh box -- must not set the figure);
    returns (median, max)."""
    import torch
    st = torch.cuda.current_stream(device).cuda_stream
    for _ in range(2):
        gf.lib.slm_gf_run(gf.h, n_frames, st)
    torch.cuda.synchronize(device)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    evs[0].record()
    for r in range(reps):
        gf.lib.slm_gf_run(gf.h, n_frames, st)
        evs[r + 1].record()
    torch.cuda.synchronize(device)
    ms = sorted(a.elapsed_time(b) for a, b in zip(evs, evs[1:]))
    return ms[len(ms) // 2], ms[-1]


def graphfit_timing(dims, device, B=8):
    """The device side of rows a18-a20 (GraphFit: 10 Adam iterations of point-plane + ARAP + Rot per frame of the bench
    workload) through the C ABI, beside cpu_baseline_autograd which times the same thing on the host: ONE frame per launch
    (what `deform_superedg` delivers to run_super.py) and B frames per launch (blockIdx.y = slot in every GraphFit kernel --
    the per-GPU share of BASELINE configs[3]), with the HBM roofline of the path on its ALGORITHMIC bytes (per
    frame and optimiser iteration the surfel pass streams 72 N bytes -- SURVEY 8d's loss-pass figure: xyz, KNN ids and
    weights, four target taps -- and the node pass reads / writes dv, grad, m1, m2 and the node KNN table: 440 J bytes)."""
    import torch
    from super_amd import synth
    from super_amd.deform_mesh import GraphFit
    opt = synth.graphfit_options(optimizer="Adam")
    it = int(opt.num_optimize_iterations)
    scs = [synth.make_scene(seed=s, **dims) for s in range(B)]
    out = {}
    for n in (1, B):
        gf = GraphFit(opt, max_frames=n)
        keep = [gf._bind(i, *_reorder(_graphfit_frames(scs[i], device))) for i in range(n)]
        ms, ms_max = _time_gf_run(gf, n, device)
        alg = n * it * (72.0 * scs[0].N + 440.0 * scs[0].J)
        gbs = alg / ms / 1e6
        # (PMC bytes of one slm_gf_run when the committed summary is of these sources, the workload C2 and B = 8)
        traffic, tsrc = rows_traffic("b1" if n == 1 else "b8") if (dims == synth.WORKLOADS["C2"] and B == 8) else (None, None)
        out[f"b{n}"] = {"ms_per_launch": ms, "ms_per_launch_max": ms_max, "ms_per_frame": ms / n, "value": n * it / (ms * 1e-3), "unit": "Adam it/s",
                        "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                                     "traffic": traffic, "traffic_source": tsrc, "algorithmic_bytes_per_launch": alg,
                                     "kernels": "one k_gf_zero, then per iteration k_gf_data (the node terms as its tail blocks) + "
                                                "k_gf_step (folds, steps, re-zeroes); one k_gf_advance"}}
        del keep, gf
    out["sample"] = (f"{it} Adam iterations per frame on {scs[0].N}-surfel / {scs[0].J}-node frames, float64 arithmetic, C ABI; "
                     f"b1 = one frame per slm_gf_run, b{B} = {B} frames per launch; HIP events on the launch stream, median of 10 runs (the slowest beside it)")
    # flat keys of rounds 1-5 (one frame per launch)
    out["ms_per_frame"], out["value"], out["unit"] = out["b1"]["ms_per_frame"], out["b1"]["value"], "Adam it/s"
    return out


def _reorder(t):
    sf, inputs, new_data = t
    return inputs, sf, new_data


def graphfit_c4_semantic(device):
    """BASELINE configs[4]'s own workload on one GPU: the Semantic-SuPer GraphFit step (soft-segmentation point-plane +
    face + boundary morphing + ARAP + Rot; reference super/deform_mesh.py:25-196,251-379) on a 500 k-surfel / 4 k-node
    frame at 720 x 960, ten SGD iterations per frame through the C ABI, float64 (the fp16-residual mode is rejected on
    evidence).  Parity of this very configuration: tests/test_gpu_edge_and_fullsize.py."""
    from super_amd import synth
    from super_amd.deform_mesh import GraphFit
    sc = synth.make_scene(seed=0, semantic=True, seg_smooth=11, **synth.WORKLOADS["C4"])
    opt = synth.graphfit_options(sf_soft_seg_point_plane=True, mesh_face=True, sf_bn_morph=True, sf_bn_morph_weight=0.1,
                                 num_classes=3)
    gf = GraphFit(opt)
    keep = gf._bind(0, *_reorder(_graphfit_frames(sc, device, semantic=True)))
    ms, ms_max = _time_gf_run(gf, 1, device, reps=5)
    it = int(opt.num_optimize_iterations)
    alg = it * (72.0 * sc.N + 440.0 * sc.J + 16.0 * sc.N)    # + the per-surfel morphing gradient (written, then read)
    del keep
    return {"ms_per_frame": ms, "ms_per_frame_max": ms_max, "value": it / (ms * 1e-3), "unit": "SGD it/s",
            "achieved_GBps_on_algorithmic_bytes": alg / ms / 1e6, "hbm_frac": alg / ms / 1e6 / HBM_PEAK_GBS,
            "boundary_pixels": list(gf.edge_counts) if gf.edge_counts else None,
            "sample": f"one {sc.N}-surfel / {sc.J}-node frame at {sc.H}x{sc.W}, 3 classes, {it} SGD iterations per frame: soft-seg "
                      "point-plane + mesh_face + sf_bn_morph + ARAP + Rot, float64, C ABI, HIP events"}


def batch_sequence(device, surfaces=8, frames=20):
    """`batch_sequence`: the headline's 8 frames per launch as a tracker of EIGHT surfaces sees them -- not the always-warm
    re-bind of the same eight problems, but eight moving surfaces at the SuPer image size (phase-shifted against each other)
    stepped in lock-step through the stage mirrors: depth preprocessing x 8 -> ONE LM_Solver.LM_batch over the eight
    models (slm_bind_frame per slot + slm_run(8)) -> update / fuseInputData / swap x 8 -> the model-side prepare of the
    next frame per slot (slm_prepare_model).  Real updates, fusion between frames, symbolic analyses and rejected
    iterations where they fall.  Reported: LM it/s over the LM stage, LM-stage and whole-step ms per frame, the share of
    (surface, frame) binds that needed a symbolic analysis, the share of rejected iterations."""
    import ctypes as C
    import numpy as np
    import torch
    from types import SimpleNamespace
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import driver_harness as drv
    from super_amd import _lib, synth
    from super_amd.LM import LM_Solver
    H, W = 480, 640
    K = synth.intrinsics()
    inv_K = np.linalg.pinv(K)
    vv, uu = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    color = torch.from_numpy(np.random.default_rng(2).uniform(0, 1, (3, H, W)).astype(np.float32))[None].to(device)

    def depth(s, k):
        d = (0.2 * synth._surface(uu, vv, H, W, 0.3 + 0.3 * np.sin(0.05 * k + 0.7 * s))).astype(np.float32)
        d[:4] = 0.0
        d[:, :4] = 0.0
        return torch.from_numpy(d)[None, None].to(device)

    opt = SimpleNamespace(height=H, width=W, data="superv1", load_valid_mask=False, depth_model="monodepth2",
                          dilate_invalid_kernel=0, normal_model="naive", phase="test", method="super", load_depth=True,
                          deform_udpate_method="super_edg", mesh_step_size=11, use_derived_gradient=True,
                          sf_point_plane=True, mesh_arap=True, mesh_rot=True, mesh_face=False, sf_point_plane_weight=1.0,
                          mesh_arap_weight=10.0, mesh_rot_weight=1.0, mesh_face_weight=1.0, num_optimize_iterations=10,
                          num_neighbors=4, num_ED_neighbors=4, th_dist=0.02, th_cosine_ang=0.4, th_time_steps=30,
                          disable_merging_new_surfels=False, disable_merging_exist_surfels=False,
                          disable_adding_new_surfels=False, disable_removing_unstable_surfels=False, slm_prepare_ahead=False)
    loops = [drv.FrameLoop(opt) for _ in range(surfaces)]      # per-surface model state (their own solvers stay unused)
    lm = LM_Solver(opt, max_frames=surfaces)
    lib = _lib.load()
    cnt = (C.c_int64 * 4)()

    def inputs_of(s, k):
        return {("depth", 0): depth(s, k), ("disp", 0): torch.zeros(1, 1, H, W, device=device),
                "inv_K": torch.from_numpy(inv_K)[None].to(device), "K": torch.from_numpy(K)[None].to(device),
                ("color", 0): color, "divterm": torch.tensor(1.0 / (2 * 0.6 * 0.6)), "filename": ["%06d" % k], "time": k,
                "ID": torch.tensor([k])}

    for s, loop in enumerate(loops):                            # frame 0: the models
        loop(SimpleNamespace(), inputs_of(s, 0))
        lm.prepare_model(loop.sf, slot=s)
    lm_ms, step_ms, builds, rej, its = [], [], [], 0, 0
    for k in range(1, frames):
        ins = [inputs_of(s, k) for s in range(surfaces)]
        torch.cuda.synchronize(device)
        lib.slm_debug_counters(cnt)
        b0 = cnt[2]
        t0 = time.perf_counter()
        tg = []
        for s, loop in enumerate(loops):
            target, ins[s] = drv.depth_preprocessing(opt, None, loop._to_device(ins[s]))
            tg.append(target)
        torch.cuda.synchronize(device)
        t1 = time.perf_counter()
        betas = lm.LM_batch([(loop.sf, ins[s], tg[s]) for s, loop in enumerate(loops)])
        torch.cuda.synchronize(device)
        t2 = time.perf_counter()
        for s, loop in enumerate(loops):
            loop.sf.update(betas[s])
            loop.sf.fuseInputData(ins[s], tg[s])
            loop.sf.prepareStableIndexNSwapAllModel(ins[s], tg[s])
            lm.prepare_model(loop.sf, slot=s)
        torch.cuda.synchronize(device)
        t3 = time.perf_counter()
        lib.slm_debug_counters(cnt)
        lm_ms.append((t2 - t1) * 1e3)
        step_ms.append((t3 - t0) * 1e3)
        builds.append(int(cnt[2] - b0))
        for recs in lm.last_records:
            its += len(recs)
            rej += sum(1 for r in recs if r["status"] == 0 and not r["accepted"])
    skip = 3
    lmv, stv = np.array(lm_ms[skip:]), np.array(step_ms[skip:])
    n_it = int(opt.num_optimize_iterations)
    surf = [int(l.sf.points.shape[0]) for l in loops]
    nodes = int(loops[0].sf.ED_nodes.num)
    return {"value": surfaces * n_it / (float(np.median(lmv)) * 1e-3), "unit": "LM it/s (LM stage, median step)",
            "lm_stage_ms_per_frame": float(np.median(lmv)) / surfaces, "lm_stage_ms": {"median": float(np.median(lmv)), "p99": float(np.percentile(lmv, 99)), "mean": float(lmv.mean())},
            "step_ms_per_frame": float(np.median(stv)) / surfaces,
            "cold_bind_share": float(sum(builds[skip:])) / (surfaces * len(builds[skip:])),
            "rejected_iteration_share": rej / max(its, 1),
            "sample": f"{surfaces} moving synthetic surfaces in lock-step, {frames - 1 - skip} counted frames each, {H}x{W}, "
                      f"{min(surf)}..{max(surf)} surfels / {nodes} nodes at the end, {n_it} LM iterations per frame, one slm_run over "
                      f"{surfaces} slots per step (hybrid solver); model-side prepare queued per slot after the swap; whole step = depth "
                      "preprocessing + LM + update + fusion + swap of all surfaces, host-timed around stream syncs"}


def next_row_timings(device):
    """Untimed-region extras (rank 0, N=1): the widened rows of SURVEY.md 8(f), measured with torch
    events on the stream the kernels run on.  f2 = depth_preprocessing at the SuPer image size."""
    import numpy as np
    import torch
    from types import SimpleNamespace
    from super_amd import synth
    from super_amd.data_loader import depth_preprocessing
    H, W = 480, 640
    K = synth.intrinsics()
    vv, uu = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    depth = torch.from_numpy((0.2 * synth._surface(uu, vv, H, W, 0.3)).astype(np.float32))[None, None].to(device)
    color = torch.rand(1, 3, H, W, device=device) * 255.0
    opt = SimpleNamespace(height=H, width=W, data="superv1", load_valid_mask=False, depth_model="monodepth2",
                          dilate_invalid_kernel=0, normal_model="naive", phase="test")
    inputs = {"inv_K": torch.from_numpy(np.linalg.pinv(K))[None], "K": torch.from_numpy(K)[None],
              ("color", 0): color, "divterm": 1.0 / (2 * 0.6 * 0.6), "filename": ["000001"]}
    reps = 20
    t = []
    for i in range(reps + 3):
        inputs[("depth", 0)] = depth.clone()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        data, _ = depth_preprocessing(opt, None, inputs)
        e1.record()
        torch.cuda.synchronize()
        if i >= 3:
            t.append(e0.elapsed_time(e1))
    ms = sum(t) / len(t)
    T = int(data.points.shape[0])
    # algorithmic bytes: depth + colour in; points/normals/colours f32, radii f64, confs f32, index_map i32, valid u8 out
    nbytes = H * W * (4 + 12) + T * (12 + 12 + 12 + 8 + 4) + H * W * (4 + 1)
    out = {"depth_preprocessing": {"ms_per_frame": ms, "image": [H, W], "valid_points": T,
                                   "algorithmic_bytes": nbytes, "achieved_GBps": nbytes / ms / 1e6,
                                   "note": "mirror call incl. dtype conversions and the one host sync"}}
    out["surfel_fusion"] = fusion_timing(device)
    return out


def fusion_timing(device):
    """f1 = fuseInputData + prepareStableIndexNSwapAllModel on the C2 surfel model (200k surfels,
    2k nodes, 480x640 frame), C calls only (buffers prepared outside the timed region)."""
    import ctypes as C
    import numpy as np
    import torch
    from types import SimpleNamespace
    from super_amd import _lib, fusion, synth
    sc = synth.make_scene(seed=0, **synth.WORKLOADS["C2"])
    rng = np.random.default_rng(0)
    n, T = sc.N, sc.T
    t64 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float64)).to(device)
    t32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
    opt = SimpleNamespace(height=sc.H, width=sc.W, th_dist=0.006, th_cosine_ang=0.8, th_time_steps=30,
                          disable_merging_new_surfels=False, disable_merging_exist_surfels=False,
                          disable_adding_new_surfels=False, disable_removing_unstable_surfels=False,
                          phase="test", method="super", num_neighbors=4)
    sf = SimpleNamespace(opt=opt, points=t64(sc.sf_points), norms=t64(sc.sf_norms),
                         colors=t32(rng.uniform(0, 255, (n, 3))), radii=t64(rng.uniform(0.002, 0.004, n)),
                         confs=t32(rng.uniform(0.2, 3.0, n)), time_stamp=t32(40.0 - rng.integers(0, 45, n)),
                         isStable=torch.ones(n, dtype=torch.bool, device=device),
                         knn_indices=torch.from_numpy(sc.sf_knn_idx).to(device), knn_w=t64(sc.sf_knn_w),
                         ED_nodes=SimpleNamespace(points=t64(sc.ed_points), radii=t64(sc.ed_radii)),
                         projdata=torch.zeros(n, 2, device=device))
    inputs = {"K": torch.from_numpy(sc.K)[None], "time": 41}
    frame = dict(points=t64(sc.tgt_points), norms=t64(sc.tgt_norms), colors=t32(rng.uniform(0, 255, (T, 3))),
                 radii=t64(rng.uniform(0.002, 0.004, T)), confs=t32(rng.uniform(0.05, 1.0, T)),
                 valid=torch.from_numpy(sc.valid).to(device).to(torch.uint8),
                 index_map=torch.from_numpy(sc.index_map).to(device).to(torch.int32))
    lib = _lib.load()
    cfg = fusion._config(sf, inputs)
    cap = n + T
    h, _ = fusion._context(lib, cfg.H, cfg.W, cap, device)
    fr = _lib.SlmNewFrame()
    fr.T, fr.time = T, 41
    for k, v in frame.items():
        setattr(fr, k, v.data_ptr())
    st = torch.cuda.current_stream(device).cuda_stream
    t_fuse, t_swap, rows = [], [], (0, 0)
    for i in range(8):
        model = fusion._Model(sf, cap, device)               # fresh copy of the model (untimed)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record()
        _lib.check(lib.slm_fuse_input_data(h, C.byref(cfg), C.byref(model.c), C.byref(fr), st), "fuse")
        e[1].record()
        n_fused = int(model.c.n)
        _lib.check(lib.slm_fuse_swap_stable(h, C.byref(cfg), C.byref(model.c), 41, None, 0, None, st), "swap")
        e[2].record()
        torch.cuda.synchronize()
        if i >= 2:
            t_fuse.append(e[0].elapsed_time(e[1]))
            t_swap.append(e[1].elapsed_time(e[2]))
        rows = (n_fused, int(model.c.n))
    # bytes: every surfel row read (+ written when fused), the frame rows, 16 layer maps written + read
    row = 3 * 8 + 3 * 8 + 3 * 4 + 8 + 4 + 4 + 1 + 4 * 4 + 4 * 8 + 2 * 4
    nbytes = 2 * rows[0] * row + T * (24 + 24 + 12 + 8 + 4) + 2 * 16 * sc.H * sc.W * 4
    ms = sum(t_fuse) / len(t_fuse)
    return {"fuse_ms_per_frame": ms, "swap_ms_per_frame": sum(t_swap) / len(t_swap), "surfels_in": n,
            "surfels_after_fuse": rows[0], "surfels_after_swap": rows[1], "algorithmic_bytes": nbytes,
            "achieved_GBps": nbytes / ms / 1e6, "note": "C calls with resident buffers; one count read-back per call (2 per frame)"}


def main():
    a = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:   # one rank per GPU: do not let every rank start a host pool as wide as the machine (CFS quota)
        torch.set_num_threads(max(1, min(16, (os.cpu_count() or 16) // world)))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N "
                             "--master-addr 127.0.0.1 bench.py --gpus N ...")
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    # Test knobs (single-GPU boxes only): BENCH_SHARE_GPU=1 puts every rank on cuda:0 and
    # BENCH_DIST_BACKEND=gloo exchanges through host memory, so the multi-rank control flow
    # can be exercised where RCCL cannot run.  The driver's runs use neither.
    share = os.environ.get("BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")
    device = torch.device("cuda", 0 if share else local_rank)
    torch.cuda.set_device(device)
    # A process group exists whenever a launcher started this rank (torchrun sets RANK / MASTER_PORT), a world of ONE rank
    # included: `python -m torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` then runs the same RCCL barrier /
    # all-gather / MAX-reduce as the N-GPU job.  A plain `python bench.py` has no group and no collective.
    launched = "RANK" in os.environ and "MASTER_PORT" in os.environ
    use_dist = world > 1 or launched
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from super_amd import synth
    from super_amd.dist import all_gather_betas
    from super_amd.engine import DeviceFrame, Engine

    dims = workload_dims(a.workload)
    B = a.frames_per_gpu
    iters = 10
    # ---- synthetic frames of this rank (seeds are global frame ids) ------------------
    scenes = [synth.make_scene(seed=rank * B + i, **dims) for i in range(B)]
    pristine = [DeviceFrame.from_scene(sc, device) for sc in scenes]
    work = [DeviceFrame.from_scene(sc, device) for sc in scenes]   # updated in place per step
    N, J = scenes[0].N, scenes[0].J
    S = max(1, min(a.streams, B))
    assert B % S == 0, "--frames-per-gpu must be divisible by --streams"
    Bs = B // S
    engs = [Engine(device, max_frames=Bs, num_iterations=iters) for _ in range(S)]
    streams = [torch.cuda.Stream(device) for _ in range(S)] if S > 1 else [torch.cuda.current_stream(device)]
    eng = engs[0]
    betas = [torch.empty((J, 7), dtype=torch.float64, device=device) for _ in range(B)]
    gathered = [None]
    local = torch.empty((B, J, 7), dtype=torch.float64, device=device)

    # "same problem every step": the four state arrays Surfels.update moves are reset from the pristine copies.  That reset
    # is the harness's, not the path's: the frames' arrays are views of one allocation per field, so it is 4 copies per
    # step whatever the number of frames (32 small launches at 8 frames otherwise).
    reset_fields = ("sf_points", "sf_norms", "ed_points", "ed_norms")

    def stacked(frames, name):
        big = torch.stack([getattr(fr, name) for fr in frames])
        for i, fr in enumerate(frames):
            setattr(fr, name, big[i])
        return big

    same_shape = all(getattr(fr, nm).shape == getattr(pristine[0], nm).shape for fr in pristine for nm in reset_fields)
    if same_shape:
        pristine_big = {nm: stacked(pristine, nm) for nm in reset_fields}
        work_big = {nm: stacked(work, nm) for nm in reset_fields}

    step_events = []

    def step():
        if same_shape:
            for nm in reset_fields:                      # same problem every step
                work_big[nm].copy_(pristine_big[nm])
        else:
            for p, w in zip(pristine, work):
                for nm in reset_fields:
                    getattr(w, nm).copy_(getattr(p, nm))
        if S > 1:
            main = torch.cuda.current_stream(device)
            for st in streams:
                st.wait_stream(main)
        for k, (e, st) in enumerate(zip(engs, streams)):
            with torch.cuda.stream(st):
                if os.environ.get("BENCH_SEQ_BIND"):     # A/B knob: one slm_bind_frame per frame
                    for i in range(Bs):
                        e.bind(i, work[k * Bs + i])
                else:
                    e.bind_batch(work[k * Bs:(k + 1) * Bs])  # loss_term.prepare (LM.py:93-94), the frames concurrently
                e.run(Bs)                                # LM_Solver.LM         (LM.py:95-117)
                for i in range(Bs):
                    e.beta(i, betas[k * Bs + i])
                    e.apply_update(i, betas[k * Bs + i])  # Surfels.update   (nodes.py:193-223)
        if S > 1:
            for st in streams:
                torch.cuda.current_stream(device).wait_stream(st)
        if use_dist:                                     # end-of-frame exchange
            torch.stack(betas, out=local)
            if backend == "nccl":
                gathered[0] = all_gather_betas(local, world * B)
            else:
                gathered[0] = all_gather_betas(local.cpu(), world * B)
        ev = torch.cuda.Event(enable_timing=True)        # end of this step on the GPU timeline (no host sync)
        ev.record()
        step_events.append(ev)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(device)

    for _ in range(a.warmup):
        step()
    fence()
    if not a.no_profile:
        for e in engs:
            e.profile(True)
    del step_events[:]
    ev0 = torch.cuda.Event(enable_timing=True)
    ev0.record()
    cpu0 = os.times()
    thr0 = thread_cpu() if os.environ.get("BENCH_THREAD_CPU") else None
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    t_end = time.perf_counter()
    cpu1 = os.times()
    elapsed = t_end - t0
    # host CPU time of this rank (all its threads) per second of the timed region: what a rank costs of the node's quota
    cpu_busy = ((cpu1.user - cpu0.user) + (cpu1.system - cpu0.system)) / max(elapsed, 1e-9)
    if thr0 is not None and rank == 0:   # diagnostic: which threads of the process burn host CPU during the timed region
        thr1 = thread_cpu()
        use = sorted(((thr1[t][0] - thr0.get(t, (0.0, ""))[0], t, thr1[t][1]) for t in thr1), reverse=True)
        print("thread CPU seconds over %.2f s: " % elapsed + ", ".join(f"{nm}[{t}] {u:.2f}" for u, t, nm in use[:14] if u > 0.005), file=sys.stderr)
        for u, t, nm in use[:3]:
            for fn in ("wchan", "syscall", "stack"):
                try:
                    print(f"   [{t}] {fn}: " + open(f"/proc/self/task/{t}/{fn}").read().strip()[:300], file=sys.stderr)
                except OSError as e:
                    print(f"   [{t}] {fn}: {e}", file=sys.stderr)
        print("   all tids: " + " ".join(str(t) for t in sorted(thr1)), file=sys.stderr)
    # per-step durations on the GPU timeline: event at the end of every step (host stalls show up as GPU idle time)
    evs = [ev0] + step_events
    step_ms = sorted(a_.elapsed_time(b_) for a_, b_ in zip(evs, evs[1:]))
    prof = None
    if not a.no_profile:
        for e in engs:
            pr = e.profile_read()
            if prof is None:
                prof = pr
            else:
                for k2, v2 in pr.items():
                    prof[k2]["ms"] += v2["ms"]
                    prof[k2]["count"] += v2["count"]
            e.profile(False)
    # worst per-iteration status over all slots of this rank (0 = every iteration ran; 3 = SLM_ITER_SOLVER_TIMEOUT)
    worst_status = max(r["status"] for e in engs for i in range(Bs) for r in e.records(i))
    cpu_busy_all = cpu_busy
    per_rank_steps = None
    if use_dist:
        t = torch.tensor([elapsed, float(worst_status)], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, worst_status = float(t[0].item()), int(t[1].item())
        t = torch.tensor([cpu_busy], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        cpu_busy_all = float(t.item())
        # per-rank step durations (one extra all-gather): a first real N-GPU run must show WHICH rank is slow and how --
        # min / median / max of every rank's per-step times on its own GPU timeline, and its wall time of the timed region
        mine = torch.tensor([step_ms[0], step_ms[len(step_ms) // 2], step_ms[-1], 1e3 * (t_end - t0)], dtype=torch.float64,
                            device=device if backend == "nccl" else "cpu")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank_steps = [[round(float(x), 3) for x in r.tolist()] for r in every]
        # test hook (tests/test_gpu_eight_rank_rehearsal.py): the betas rank 0 holds after the last end-of-frame all-gather
        dump = os.environ.get("BENCH_DUMP_BETAS")
        if dump and rank == 0 and gathered[0] is not None:
            np.save(dump, gathered[0].cpu().numpy())

    recs = eng.records(0)
    final_loss = [r["loss"] for r in recs if r["status"] == 0]
    n_ok = sum(1 for r in recs if r["status"] == 0)

    if rank == 0:
        total_iters = world * B * iters * a.steps
        out = {
            "metric": "LM iterations/s (embedded-deformation LM, point-to-plane + ARAP + Rot)",
            "value": total_iters / elapsed,
            "unit": "LM it/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * elapsed / a.steps,
            "ms_per_frame": 1e3 * elapsed / (a.steps * B),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{a.workload}: {N} surfels / {J} ED nodes / KNN=4, "
                                   f"point-to-plane + ARAP + Rot, {iters} LM iterations + "
                                   f"Surfels.update per frame",
                       "frames_per_gpu": B, "streams_per_gpu": S, "global_frames": world * B,
                       "image": [dims["H"], dims["W"]],
                       "storage": "f32/i32 state in HBM (BASELINE's fp32 configs; the drop-in mirror passes the "
                                  "reference's float64 state in place), f64 arithmetic and solve",
                       "bind": "one slm_bind_frame per frame" if os.environ.get("BENCH_SEQ_BIND") else
                               "slm_bind_frames: the frames of a step bound concurrently (worker threads / streams inside the library)",
                       "parallelism": f"frames sharded over {world} GPU(s), beta all-gather"},
            "lm_iterations_ok_frame0": n_ok,
            "final_loss_frame0": final_loss[-1] if final_loss else None,
            "step_ms": {"median": step_ms[len(step_ms) // 2], "p99": step_ms[min(len(step_ms) - 1, int(0.99 * len(step_ms)))],
                        "max": step_ms[-1], "min": step_ms[0], "steps": len(step_ms)},
            "distributed": {"backend": backend if use_dist else None, "world": world,
                            "launched_by": "torchrun" if launched else None},
            "worst_iter_status_all_ranks": worst_status,
        }
        if per_rank_steps is not None:
            out["distributed"]["per_rank_step_ms"] = {"columns": ["min", "median", "max", "timed_region_wall_ms"],
                                                      "ranks": per_rank_steps}
        if prof is not None:
            info = eng.plan_info(0)
            flops = Bs * info["factor_flops"]          # padded dense-front FLOPs of one factorisation (what the MFMAs execute)
            flops_exact = Bs * info["factor_flops_unpadded"]
            sp = prof["solve"]
            savg = sp["ms"] / max(sp["count"], 1) * 1e-3
            tf_pad = flops / savg / 1e12 if savg > 0 else 0.0
            tf = flops_exact / savg / 1e12 if savg > 0 else 0.0
            solve_traffic, solve_tsrc = pmc_phase_traffic("solve", a.workload, Bs)
            # the time-dominant phase: the float64 multifrontal factor + substitutions of all frames of a launch.
            # `achieved` / `frac` are on the ALGORITHMIC FLOPs (true pivot / boundary sizes); the 64-padded count -- what
            # the MFMAs execute -- rides along as *_padded.
            out["roofline"] = {"kernel": "solve phase: " + info["solver"] + " Cholesky factor + substitutions ("
                                         + ("k_fdag, one persistent launch" if Bs <= 2 and info["solver_tasks"] > 0
                                            else "per-level launches k_fL11 / k_fL21 / k_fschur / k_fpanel / k_ftrail for the lower levels + k_fdag (task graph) for "
                                                 "the root front, its children and the back substitution of the whole tree; all launches of one LM iteration") + ")",
                               "bound": "mfma", "achieved": tf, "peak": F64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": tf / F64_MFMA_PEAK_TFLOPS,
                               "traffic": solve_traffic, "traffic_source": solve_tsrc,
                               "traffic_note": "HBM bytes of all solver launches of ONE LM iteration (PMC: 2 x FETCH_SIZE + WRITE_SIZE, separate passes)",
                               "achieved_padded": tf_pad, "frac_padded": tf_pad / F64_MFMA_PEAK_TFLOPS,
                               "avg_phase_ms": savg * 1e3, "share_of_iteration": None,
                               "factor_gflop_per_frame": info["factor_flops"] / 1e9,
                               "factor_gflop_per_frame_unpadded": info["factor_flops_unpadded"] / 1e9,
                               "fronts": int(info["fronts"]), "levels": int(info["levels"]),
                               "note": "latency-bound: a chain of dependent 64x64 float64 tile factorisations "
                                       "(33 at C2) bounds the phase, not the MFMA rate"}
            g = prof["data_grad"]
            per_launch_bytes = Bs * (72.0 * N + 3165.0 * J)    # grad pass, f32/i32
            avg_s = g["ms"] / max(g["count"], 1) * 1e-3
            ach = per_launch_bytes / avg_s / 1e9 if avg_s > 0 else 0.0
            traffic, tsrc = pmc_traffic("k_data_gram", a.workload, Bs)
            out["roofline_data"] = {"kernel": "k_data_gram", "bound": "hbm", "achieved": ach,
                                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                                    "traffic": traffic, "traffic_source": tsrc,
                                    "avg_launch_ms": avg_s * 1e3, "launches": g["count"],
                                    "algorithmic_bytes_per_launch": per_launch_bytes}
            out["plan"] = {k: (int(v) if isinstance(v, float) and k not in ("factor_flops", "factor_bytes", "factor_flops_unpadded") else v)
                           for k, v in info.items()}
            phases = {k: v["ms"] / max(v["count"], 1) for k, v in prof.items()}
            out["phase_ms_per_iteration"] = phases
            tot = sum(phases.values())
            out["roofline"]["share_of_iteration"] = phases["solve"] / tot if tot > 0 else None
            # whole step against the HBM peak: algorithmic bytes of one LM iteration x iterations x frames
            step_bytes = world * B * iters * (144.0 * N + 3205.0 * J)
            out["whole_step_hbm_frac"] = step_bytes / (elapsed / a.steps) / 1e9 / HBM_PEAK_GBS
        if world == 1 and not a.no_profile and not a.no_latency_b1:
            out["latency_b1"] = latency_b1(dims, device, iters)
            out["latency_b1_sequence"] = latency_b1_sequence(device)
            if a.workload in ("C1", "C2") and S == 1:
                out["batch_depth"] = batch_depth(scenes, device, iters)
            if a.workload == "C2" and S == 1 and B == 8:
                out["batch_sequence"] = batch_sequence(device)
        out["host"] = host_info()
        out["host"]["cpu_cores_busy_per_rank"] = round(cpu_busy, 2)   # (rank 0; timed region; user + system time of all threads)
        out["host"]["cpu_cores_busy_all_ranks"] = round(cpu_busy_all, 2)   # (sum over the ranks: what the job costs of the node's CPU quota)
        if world == 1 and not a.no_profile:
            out["bind"] = bind_timing(dims, device, B)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"], oracle_beta, oracle_trace = cpu_baseline(dims, seed=0)
            out["parity_" + a.workload.lower()] = parity_check(dims, device, oracle_beta, oracle_trace, B)
            assert out["parity_" + a.workload.lower()]["ok"], out["parity_" + a.workload.lower()]
            out["cpu_baseline_autograd"] = cpu_baseline_autograd(dims, seed=0)
            out["graphfit_gpu"] = graphfit_timing(dims, device, B)
            out["roofline_graphfit"] = out["graphfit_gpu"][f"b{B}"]["roofline"]
            out["graphfit_c4_semantic"] = graphfit_c4_semantic(device)
            out["next_rows"] = next_row_timings(device)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
