"""Diagnostic: 8 C2 frames bound one by one (slm_bind_frame) vs concurrently (slm_bind_frames), plans cached."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-super_amd")); sys.path.insert(0, ROOT)
import torch
from super_amd import synth
from super_amd.engine import DeviceFrame, Engine
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
frames = [DeviceFrame.from_scene(synth.make_scene(seed=s, **synth.WORKLOADS["C2"]), dev) for s in range(B)]
eng = Engine(dev, max_frames=B)
for rep in range(2):
    for i, f in enumerate(frames):
        eng.bind(i, f)
    eng.bind_batch(frames)
torch.cuda.synchronize()
for name, fn in (("one by one", lambda: [eng.bind(i, f) for i, f in enumerate(frames)]), ("slm_bind_frames", lambda: eng.bind_batch(frames))):
    ts = []
    for rep in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        ts.append(((t1 - t0) * 1e3, (time.perf_counter() - t0) * 1e3))
    print(f"{name:16s}: host {min(t[0] for t in ts):.3f} ms, to completion {min(t[1] for t in ts):.3f} ms for {B} frames; all: {[round(t[1], 2) for t in ts]}")
# does the LM run after a batch bind take as long as after one-by-one binds?
for name, fn in (("one by one", lambda: [eng.bind(i, f) for i, f in enumerate(frames)]), ("slm_bind_frames", lambda: eng.bind_batch(frames))) * 2:
    ts = []
    for rep in range(4):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.run(B)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"run after {name:16s}: {min(ts):.2f} ms  all {[round(t, 2) for t in ts]}")
# bind + run back to back (no sync in between), on the default stream and on a side stream
for sname, stream in (("null stream", None), ("side stream", torch.cuda.Stream(dev))):
    for name, fn in (("one by one", lambda: [eng.bind(i, f) for i, f in enumerate(frames)]), ("slm_bind_frames", lambda: eng.bind_batch(frames))) * 2:
        ts = []
        for rep in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if stream is None:
                fn(); eng.run(B)
            else:
                with torch.cuda.stream(stream):
                    fn(); eng.run(B)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print(f"{sname}: bind+run {name:16s}: {min(ts):.2f} ms  all {[round(t, 2) for t in ts]}")
# the whole bench step: restore the state, bind, run, beta, update
import copy
pristine = [DeviceFrame.from_scene(synth.make_scene(seed=s, **synth.WORKLOADS["C2"]), dev) for s in range(B)]
betas = [torch.empty((f.J, 7), dtype=torch.float64, device=dev) for f in frames]
def full_step(batch, restore=True, update=True):
    if restore:
        for p, w in zip(pristine, frames):
            w.sf_points.copy_(p.sf_points); w.sf_norms.copy_(p.sf_norms); w.ed_points.copy_(p.ed_points); w.ed_norms.copy_(p.ed_norms)
    if batch:
        eng.bind_batch(frames)
    else:
        for i, f in enumerate(frames):
            eng.bind(i, f)
    eng.run(B)
    for i in range(B):
        eng.beta(i, betas[i])
        if update:
            eng.apply_update(i, betas[i])
for restore, update in ((True, True), (False, True), (True, False), (False, False)):
    for batch in (False, True, False, True):
        full_step(batch, restore, update); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for rep in range(4):
            full_step(batch, restore, update)
        torch.cuda.synchronize()
        print(f"step restore={restore} update={update} batch_bind={batch}: {(time.perf_counter() - t0) * 250:.2f} ms")
