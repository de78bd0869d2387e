"""Diagnostic: per-task timeline of the task-graph solver (solver_path 2) for one damped solve.
    python tools/trace_dag.py [workload] [frames]"""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "python-super_amd")); sys.path.insert(0, ROOT)
import numpy as np
import torch
from super_amd import _lib, synth
from super_amd.engine import DeviceFrame, Engine
dev = torch.device("cuda", 0)
wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
HYBRID = "--hybrid" in sys.argv      # the top-of-tree task list of the hybrid form (solver_path 4) instead of the whole tree
e = Engine(dev, solver_path=4 if HYBRID else 2, max_frames=B)
for i in range(B):
    e.bind(i, DeviceFrame.from_scene(synth.make_scene(seed=i, **synth.WORKLOADS[wl]), dev))
e.run(B)
torch.cuda.synchronize()
_lib.check(e.lib.slm_debug_dag_trace(e.h, 0, 1, e.stream), "trace on")
e.run(B)
torch.cuda.synchronize()


def read(what, dtype):
    n = C.c_int64(0)
    _lib.check(e.lib.slm_debug_read(e.h, 0, what, None, 0, C.byref(n), e.stream), "dbg")
    a = np.zeros(n.value)
    _lib.check(e.lib.slm_debug_read(e.h, 0, what, a.ctypes.data_as(C.c_void_p), n.value, C.byref(n), e.stream), "dbg")
    return a.view(dtype)


tr = read(4, np.int64).reshape(-1, 24)
tk = read(6 if HYBRID else 5, np.int32).reshape(-1, 2)
tr = tr[:len(tk)]
typ, front, r, s = tk[:, 0] >> 24, tk[:, 0] & 0xFFFFFF, tk[:, 1] >> 8, tk[:, 1] & 255
t0 = tr[:, 0].min()
st, rd, en = (tr[:, 0] - t0) / 100.0, (tr[:, 1] - t0) / 100.0, (tr[:, 2] - t0) / 100.0   # microseconds
names = ["POTRF", "COL", "SCHUR", "BACKB", "BACK"]
print(f"{wl} B={B}: {len(tk)} tasks (slot 0), span {en.max():.1f} us, workgroups used {len(np.unique(tr[:, 3]))}")
for t in range(5):
    m = typ == t
    if m.any():
        print(f"  {names[t]:6s} x{m.sum():5d}: start..end mean {np.mean(en[m] - st[m]):7.2f} us, waiting {np.mean(rd[m] - st[m]):7.2f}, "
              f"after last dependency {np.mean(en[m] - rd[m]):6.2f} (max {np.max(en[m] - rd[m]):6.2f}); busy sum {np.sum(en[m] - rd[m]) / 1e3:.2f} ms")
fact_end = en[typ <= 2].max()
print(f"  factorisation + forward done at {fact_end:.1f} us; back substitution takes {en.max() - fact_end:.1f} us")
# the chain of the root front
root = front.max()
print("root front timeline (us):")
for i in np.argsort(st):
    if front[i] == root and typ[i] in (0, 4) or (front[i] == root and typ[i] == 1 and r[i] == s[i] + 1):
        print(f"   {names[typ[i]]:6s} r={r[i]:2d} s={s[i]:2d}  start {st[i]:8.1f}  ready {rd[i]:8.1f}  end {en[i]:8.1f}  wg {tr[i, 3]}")
m = (typ == 0) & (front == root) & (s > 0)
if m.any():
    sub = (tr[m][:, [1, 4, 5, 6, 7, 2]] - tr[m][:, [1]]) / 100.0
    print("root POTRF sub-steps after the last dependency (us): accumulate, reduce + tile to LDS, factor + inverse, stores + y, publish:")
    print("   ", np.round(np.diff(sub, axis=1).mean(0), 2))
    ft = (tr[m][:, 8:21] - tr[m][:, [8]]) / 100.0
    print("inside the factorisation (us from its start): per 16 pivots [start, diag16 done, barrier passed] x 4, end:")
    print("   ", np.round(ft.mean(0), 2))
# per level: when does the first / last POTRF of each depth finish
order = np.argsort(en)
lvl_front = {}
for i in order:
    if typ[i] == 0:
        lvl_front.setdefault(front[i], []).append((s[i], rd[i], en[i]))
ends = sorted((max(x[2] for x in v), f, len(v)) for f, v in lvl_front.items())
print("last 12 fronts to finish factoring (end us, front, npt):", [(round(a, 1), int(b), c) for a, b, c in ends[-12:]])
if "--chain" in sys.argv:
    # absolute stamps (us) of the root front's POTRF tasks: where does the hand-off between two columns go?
    print("root chain, absolute us: s | start ready | mark4 mark5 mark6 mark7 end | factor stamps (start, [diag16 done, barrier] x 4, end)")
    for i in sorted(np.nonzero((typ == 0) & (front == root))[0], key=lambda i: s[i]):
        a = (tr[i] - t0) / 100.0
        print(f"  s={s[i]}  {a[0]:8.2f} {a[1]:8.2f} | " + " ".join(f"{x:8.2f}" for x in a[4:8]) + f" {a[2]:8.2f} | " + " ".join(f"{x:7.2f}" for x in a[8:21]))
if "--transition" in sys.argv:
    # from the last pivot column of the root's children to the root's first factorisation
    kids = sorted(set(front[(typ == 2)]))[-2:]          # the two fronts with SCHUR tasks that finish last = the root's children
    print("transition into the root front (us):")
    for k in kids:
        mp = (front == k) & (typ == 0)
        last = np.argmax(np.where(mp, en, -1))
        print(f"  child {k}: npt {int(s[mp].max()) + 1}; last POTRF s={s[last]} ready {rd[last]:.1f} end {en[last]:.1f};"
              f" its factor {(tr[last, 8] - t0) / 100:.1f} .. {(tr[last, 20] - t0) / 100:.1f}")
        mc = (front == k) & (typ == 1) & (s == s[last])
        print(f"     COL tasks of that column: {mc.sum()}, ready {rd[mc].min():.1f} .. {rd[mc].max():.1f}, end {en[mc].min():.1f} .. {en[mc].max():.1f}")
        ms = (front == k) & (typ == 2)
        print(f"     SCHUR tasks: {ms.sum()}, ready {rd[ms].min():.1f} .. {rd[ms].max():.1f}, end {en[ms].min():.1f} .. {en[ms].max():.1f}")
    m0 = (front == root) & (typ == 0) & (s == 0)
    i0 = np.nonzero(m0)[0][0]
    a = (tr[i0] - t0) / 100.0
    print(f"  root POTRF(0): start {a[0]:.1f}, stage 0 done (mark4) {a[4]:.1f}, factor {a[8]:.1f} .. {a[20]:.1f}")
    mc0 = (front == root) & (typ == 1) & (s == 0)
    print(f"  root COL(r,0): ready {rd[mc0].min():.1f} .. {rd[mc0].max():.1f}, end {en[mc0].min():.1f} .. {en[mc0].max():.1f}")
if "--kids" in sys.argv:
    # the chains of the root's two children (the fronts below the root with the most pivot columns): per POTRF start / ready / end,
    # and when the COL tasks of each column ran -- at 8 frames per launch these chains compete for workgroups with 7 other slots
    npt_of = {f: len(v) for f, v in lvl_front.items()}
    kids = sorted((f for f in npt_of if f != root), key=lambda f: -npt_of[f])[:int(os.environ.get("TRACE_KIDS", "2"))]   # (TRACE_KIDS=6: the level below too)
    for k in kids:
        print(f"child front {k} (npt {npt_of[k]}): s | POTRF start ready end wg | COL tasks: n, first ready, last end")
        for sc in range(npt_of[k]):
            i = np.nonzero((typ == 0) & (front == k) & (s == sc))[0][0]
            mc = (typ == 1) & (front == k) & (s == sc)
            extra = f"{mc.sum():3d} {rd[mc].min():8.1f} {en[mc].max():8.1f} (start {st[mc].min():.1f} .. {st[mc].max():.1f})" if mc.any() else ""
            print(f"   s={sc}  {st[i]:8.1f} {rd[i]:8.1f} {en[i]:8.1f} wg {tr[i, 3]:3d} | {extra}")

if "--schur" in sys.argv:
    # the update-matrix tasks per front: when were they taken, when ready, when done (a level transition = the last of them + the parent's gather)
    for k in sorted(set(front[typ == 2])):
        m = (typ == 2) & (front == k)
        print(f"SCHUR of front {k}: {m.sum():3d} tasks, start {st[m].min():7.1f} .. {st[m].max():7.1f}, ready {rd[m].min():7.1f} .. {rd[m].max():7.1f}, "
              f"end {en[m].min():7.1f} .. {en[m].max():7.1f}, busy sum {np.sum(en[m] - rd[m]):7.1f} us, mean {np.mean(en[m] - rd[m]):5.1f}")
    # how many workgroups hold a task at time t / work on one (past their last dependency)
    for t in range(0, int(en.max()) + 1, 20):
        held = int(((st <= t) & (en > t)).sum())
        work = int(((rd <= t) & (en > t)).sum())
        print(f"  t = {t:4d} us: {held:3d} tasks held, {work:3d} past their last dependency")

if "--fronts" in sys.argv:
    # per front: pivot tile columns, when its first chain task was taken / ready, when its last one ended, when its last update task ended
    print("front npt | first POTRF taken, ready | last POTRF end | us per column | last SCHUR end | COL tasks taken late by (mean, max us)")
    for k in sorted(set(front[typ == 0]), key=lambda k: en[(typ == 0) & (front == k)].max()):
        mp = (typ == 0) & (front == k)
        i0 = np.nonzero(mp & (s == 0))[0][0]
        ms = (typ == 2) & (front == k)
        mc = (typ == 1) & (front == k)
        npt = int(s[mp].max()) + 1
        # a COL task is "late" by how long after its column's POTRF ended it was taken (<= 0: it was waiting already)
        late = np.array([st[i] - en[np.nonzero(mp & (s == s[i]))[0][0]] for i in np.nonzero(mc)[0]]) if mc.any() else np.zeros(1)
        print(f"  {k:4d} {npt:2d} | {st[i0]:7.1f} {rd[i0]:7.1f} | {en[mp].max():7.1f} | {(en[mp].max() - rd[i0]) / npt:5.1f} | "
              f"{(en[ms].max() if ms.any() else float('nan')):7.1f} | {np.maximum(late, 0).mean():5.1f} {late.max():6.1f}")
