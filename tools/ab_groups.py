"""Diagnostic (GPU box): A/B of the grouped run (two phase-shifted groups of frames inside ONE slm_run, the task-graph
launch of each group capped to a fraction of the CUs) against the single-group hybrid form, on ONE box.

    python tools/ab_groups.py [workload=C2] [frames=8] [reps=7]

Every variant is a fresh solver created under its own environment knobs (read in slm_create); the same frames are bound
to all of them, and the final beta of every slot is compared with the single-group run (data_path 2 variants: bitwise).
Prints ms per LM iteration (bind excluded), the solve-phase time of the profile events (summed over the groups) and
it/s."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-super_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
from super_amd import synth
from super_amd.engine import DeviceFrame, Engine

dev = torch.device("cuda", 0)
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 7

VARIANTS = [("single", {})]
for G, caps in ((2, (128, 160, 192, 0)), (3, (96, 128)), (4, (64, 96, 128))):
    for cap in caps:
        VARIANTS.append((f"G{G} free cap{cap}", {"SLM_GROUPS": str(G), "SLM_DAG_CAP": str(cap), "SLM_GROUP_SYNC": "0"}))
for cap in (128, 192):
    VARIANTS.append((f"G2 stagger cap{cap}", {"SLM_GROUPS": "2", "SLM_DAG_CAP": str(cap), "SLM_GROUP_SYNC": "2"}))
VARIANTS.append(("G2 sync cap96", {"SLM_GROUPS": "2", "SLM_DAG_CAP": "96", "SLM_GROUP_SYNC": "1"}))
VARIANTS.append(("single (again)", {}))
if os.environ.get("AB_VARIANTS"):
    keep = os.environ["AB_VARIANTS"].split(",")
    VARIANTS = [v for v in VARIANTS if any(k in v[0] for k in keep)]

KNOBS = ("SLM_GROUPS", "SLM_DAG_CAP", "SLM_GROUP_SYNC", "SLM_GROUP_MIN")
frames = [DeviceFrame.from_scene(synth.make_scene(seed=s, **synth.WORKLOADS[wl]), dev) for s in range(B)]


def run_variant(env, data_path=0):
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    eng = Engine(dev, max_frames=B, data_path=data_path)
    eng.bind_batch(frames)
    eng.run(B)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        eng.bind_batch(frames)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run(B)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    betas = [eng.beta(i).cpu().numpy() for i in range(B)]
    recs = [eng.records(i) for i in range(B)]
    eng.bind_batch(frames)
    eng.profile(True)
    eng.run(B)
    ph = eng.profile_read()
    eng.profile(False)
    eng.close()
    phases = {k: v["ms"] / max(v["count"], 1) for k, v in ph.items()}
    return min(ts) / 10, sorted(ts)[len(ts) // 2] / 10, phases, betas, recs


base = None
for name, env in VARIANTS:
    best, med, phases, betas, recs = run_variant(env)
    if base is None:
        base = (betas, recs)
    err = max(float(np.abs(a - b).max()) for a, b in zip(betas, base[0]))
    same_flags = all([r["accepted"] for r in ra] == [r["accepted"] for r in rb] for ra, rb in zip(recs, base[1]))
    print(f"{wl} B={B} {name:24s}: {best:.3f} ms/iter best, {med:.3f} median ({10 * B / (10 * best) * 1e3:.0f} it/s); "
          f"solve {phases['solve']:.3f} data_grad {phases['data_grad']:.3f} zero {phases['zero']:.3f} "
          f"reg {phases['reg_grad']:.3f} loss {phases['data_loss']:.3f} accept {phases['accept']:.3f}; "
          f"max|beta - single| {err:.2e} flags_equal {same_flags}", flush=True)

# bitwise check on the run-to-run reproducible data path: a frame sees the launches of a batch of its GROUP's size
if B % 2 == 0 and not os.environ.get("AB_VARIANTS"):
    b1 = run_variant({"SLM_GROUPS": "2", "SLM_DAG_CAP": "128", "SLM_GROUP_SYNC": "0"}, data_path=2)[3]
    half = B // 2
    for k in KNOBS:
        os.environ.pop(k, None)
    ok = True
    for g in range(2):
        eng = Engine(dev, max_frames=half, data_path=2)
        eng.bind_batch(frames[g * half:(g + 1) * half])
        eng.run(half)
        ok = ok and all(np.array_equal(eng.beta(i).cpu().numpy(), b1[g * half + i]) for i in range(half))
        eng.close()
    print("data_path 2: grouped beta bitwise equal to the two half batches run one after the other:", ok, flush=True)
