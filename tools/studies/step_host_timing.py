"""Study: host-side and device-side time of the pieces of one bench step (C2, 8 frames): reset copies, bind_batch, run,
beta + apply_update.  Each piece is followed by a device synchronize here (unlike bench.py), so the sum is larger than
a pipelined step; the point is which piece carries the milliseconds outside the LM iterations."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, os.path.join(ROOT, "python-super_amd"))
from super_amd import synth  # noqa: E402
from super_amd.engine import DeviceFrame, Engine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
scenes = [synth.make_scene(seed=i, **synth.WORKLOADS["C2"]) for i in range(B)]
pristine = [DeviceFrame.from_scene(sc, dev) for sc in scenes]
work = [DeviceFrame.from_scene(sc, dev) for sc in scenes]
eng = Engine(dev, max_frames=B, num_iterations=10)
J = scenes[0].J
betas = [torch.empty((J, 7), dtype=torch.float64, device=dev) for _ in range(B)]
names = ["reset", "bind_batch", "run", "beta+update"]
acc = {n: [] for n in names}


def timed(name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    acc[name].append((1e3 * (t1 - t0), 1e3 * (t2 - t0)))


def reset():
    for p, w in zip(pristine, work):
        for nm in ("sf_points", "sf_norms", "ed_points", "ed_norms"):
            getattr(w, nm).copy_(getattr(p, nm))


def upd():
    for i in range(B):
        eng.beta(i, betas[i])
        eng.apply_update(i, betas[i])


for it in range(8):
    timed("reset", reset)
    timed("bind_batch", lambda: eng.bind_batch(work))
    timed("run", lambda: eng.run(B))
    timed("beta+update", upd)
for n in names:
    a = np.array(acc[n][2:])
    print(f"{n:12s} host enqueue {a[:, 0].mean():7.3f} ms   until the device is idle {a[:, 1].mean():7.3f} ms   (min {a[:, 1].min():.3f}, max {a[:, 1].max():.3f})")

# ---- pipelined, as bench.py runs it: no synchronisation between the pieces; host time stamps only ----
print("pipelined steps (ms): step time | bind_batch call (includes the wait for the previous run) | run enqueue | reset+update enqueue")
torch.cuda.synchronize()
prev = time.perf_counter()
rows = []
for it in range(14):
    t0 = time.perf_counter()
    reset()
    t1 = time.perf_counter()
    eng.bind_batch(work)
    t2 = time.perf_counter()
    eng.run(B)
    t3 = time.perf_counter()
    upd()
    t4 = time.perf_counter()
    rows.append((t0, t1, t2, t3, t4))
torch.cuda.synchronize()
tend = time.perf_counter()
for k in range(2, len(rows)):
    t0, t1, t2, t3, t4 = rows[k]
    nxt = rows[k + 1][0] if k + 1 < len(rows) else tend
    # a step "ends" when its successor's bind returns (the device finished its run); use bind-return to bind-return
    print(f"  step {k:2d}: bind-return to bind-return {1e3 * ((rows[k + 1][2] if k + 1 < len(rows) else tend) - t2):7.3f} | bind call {1e3 * (t2 - t1):7.3f} | run enqueue {1e3 * (t3 - t2):6.3f} | "
          f"reset {1e3 * (t1 - t0):6.3f} update {1e3 * (t4 - t3):6.3f}")


def cpu_stat():
    try:
        return {ln.split()[0]: int(ln.split()[1]) for ln in open("/sys/fs/cgroup/cpu.stat")}
    except OSError:
        return {}


# ---- 120 pipelined steps: how often does a step stall, and is the process being throttled (cgroup cpu.stat)? ----
s0 = cpu_stat()
torch.cuda.synchronize()
marks = []
for it in range(120):
    reset()
    eng.bind_batch(work)
    marks.append(time.perf_counter())
    eng.run(B)
    upd()
torch.cuda.synchronize()
s1 = cpu_stat()
d = 1e3 * np.diff(np.array(marks))
print(f"120 steps: median {np.median(d):.3f} ms, mean {d.mean():.3f}, p90 {np.percentile(d, 90):.3f}, max {d.max():.3f}; steps > median + 1 ms: {(d > np.median(d) + 1).sum()}")
print("   slow steps (index: ms):", {int(i): round(float(d[i]), 2) for i in np.nonzero(d > np.median(d) + 1)[0]})
print("   cgroup cpu.stat deltas:", {k: s1[k] - s0[k] for k in s0 if k in s1 and s1[k] != s0[k]})
