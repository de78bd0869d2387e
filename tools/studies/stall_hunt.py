"""Study: the rare ~8 ms stall of a bench step.

    python tools/studies/stall_hunt.py run [steps=200] [frames=8]        # per-step / per-segment timing, outliers
    rocprofv3 --hip-trace --kernel-trace --output-format csv -d /tmp/sh -o sh -- python3 tools/studies/stall_hunt.py run 200
    python tools/studies/stall_hunt.py gaps "/tmp/sh/**/" [gap_ms=2]      # GPU idle gaps and the HIP calls spanning them

`run` executes bench.py's step (reset copies, slm_bind_frames, slm_run, beta + Surfels.update per frame) with a device
event after every segment and host time stamps around every call: for an outlier step it tells whether the GPU was
busy longer (a slow kernel) or idle (the host did not feed it), and in which segment.  `gaps` reads a rocprofv3 trace of
the same run: every idle gap of the GPU timeline longer than gap_ms with the kernels either side of it and the HIP API
calls that were in progress during the gap, longest first."""
import csv
import glob
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "python-super_amd"))
sys.path.insert(0, ROOT)


def run(steps, B):
    import gc
    import numpy as np
    import torch
    from super_amd import synth
    from super_amd.engine import DeviceFrame, Engine
    dev = torch.device("cuda", 0)
    scenes = [synth.make_scene(seed=i, **synth.WORKLOADS["C2"]) for i in range(B)]
    pristine = [DeviceFrame.from_scene(sc, dev) for sc in scenes]
    work = [DeviceFrame.from_scene(sc, dev) for sc in scenes]
    eng = Engine(dev, max_frames=B, num_iterations=10)
    J = scenes[0].J
    betas = [torch.empty((J, 7), dtype=torch.float64, device=dev) for _ in range(B)]
    fields = ("sf_points", "sf_norms", "ed_points", "ed_norms")
    seg_names = ["reset", "bind", "run", "update"]
    recs = []
    gc_events = []
    gc.callbacks.append(lambda phase, info: gc_events.append((time.perf_counter(), phase, info.get("generation"))))

    def ev():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    for s in range(steps + 3):
        host = [time.perf_counter()]
        evs = [ev()]
        for p, w in zip(pristine, work):
            for nm in fields:
                getattr(w, nm).copy_(getattr(p, nm))
        host.append(time.perf_counter()); evs.append(ev())
        eng.bind_batch(work)
        host.append(time.perf_counter()); evs.append(ev())
        eng.run(B)
        host.append(time.perf_counter()); evs.append(ev())
        for i in range(B):
            eng.beta(i, betas[i])
            eng.apply_update(i, betas[i])
        host.append(time.perf_counter()); evs.append(ev())
        recs.append((host, evs))
    torch.cuda.synchronize()
    recs = recs[3:]
    gpu = np.array([[a.elapsed_time(b) for a, b in zip(e, e[1:])] for _, e in recs])          # (steps, 4) ms on the GPU timeline
    hst = np.array([[1e3 * (b - a) for a, b in zip(h, h[1:])] for h, _ in recs])              # host time inside each call
    step_gpu = np.array([recs[i][1][0].elapsed_time(recs[i + 1][1][0]) for i in range(len(recs) - 1)])
    med = np.median(step_gpu)
    print(f"{len(step_gpu)} steps of {B} C2 frames: median {med:.2f} ms, p99 {np.percentile(step_gpu, 99):.2f}, max {step_gpu.max():.2f}")
    print("segment medians (GPU timeline ms | host ms inside the call): " +
          ", ".join(f"{n} {np.median(gpu[:, k]):.2f} | {np.median(hst[:, k]):.2f}" for k, n in enumerate(seg_names)))
    out = np.nonzero(step_gpu > med + 2.0)[0]
    print(f"outlier steps (> median + 2 ms): {[int(i) for i in out]}")
    if len(out) > 2:
        ts = np.array([recs[i][0][0] - recs[0][0][0] for i in out])
        print("   wall-clock spacing of the outliers (s): " + " ".join(f"{d:.3f}" for d in np.diff(ts)))
    t_first = recs[0][0][0]
    for i in out:
        print(f"  step {i} at t={recs[i][0][0] - t_first:8.3f} s: {step_gpu[i]:.2f} ms;  GPU segments " + ", ".join(f"{n} {gpu[i, k]:.2f}" for k, n in enumerate(seg_names)) +
              ";  host " + ", ".join(f"{n} {hst[i, k]:.2f}" for k, n in enumerate(seg_names)))
        a, b = recs[i][0][0], recs[i + 1][0][0]
        g = [(t - a, ph, gen) for t, ph, gen in gc_events if a <= t <= b]
        if g:
            print("     Python gc during the step: " + ", ".join(f"{ph} gen{gen} at +{1e3 * t:.2f} ms" for t, ph, gen in g))
    try:
        st = open("/sys/fs/cgroup/cpu.stat").read().split()
        print("cgroup cpu.stat:", " ".join(st))
    except OSError:
        pass


def gaps(pattern, gap_ms):
    kt = glob.glob(os.path.join(pattern, "*kernel_trace.csv"), recursive=True)
    ht = glob.glob(os.path.join(pattern, "*hip_api_trace.csv"), recursive=True)
    rows = list(csv.DictReader(open(kt[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    api = list(csv.DictReader(open(ht[0]))) if ht else []
    name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
    prev_end, prev = int(rows[0]["Start_Timestamp"]), None
    t0 = prev_end
    found = 0
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if (s - prev_end) / 1e6 > gap_ms:
            found += 1
            print(f"t={(s - t0) / 1e6:9.2f} ms  GPU idle {(s - prev_end) / 1e6:7.2f} ms  after {name(prev) if prev else '-'}  before {name(r)}")
            spans = []
            for a in api:
                a0, a1 = int(a["Start_Timestamp"]), int(a["End_Timestamp"])
                if a1 > prev_end and a0 < s and (a1 - a0) > 50_000:
                    spans.append(((a1 - a0) / 1e6, a["Function"], (a0 - prev_end) / 1e6, a.get("Thread_Id", "?")))
            for d, fn, off, tid in sorted(spans, reverse=True)[:8]:
                print(f"      {fn:32s} {d:8.3f} ms  (starts {off:+.3f} ms from the gap's start, thread {tid})")
        if e > prev_end:
            prev_end, prev = e, r
    print(f"{len(rows)} kernels, {found} gaps > {gap_ms} ms")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "gaps":
        gaps(sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 2.0)
    else:
        a = [x for x in sys.argv[1:] if x != "run"]
        run(int(a[0]) if a else 200, int(a[1]) if len(a) > 1 else 8)
