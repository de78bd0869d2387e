"""Diagnostic: per-LAUNCH durations of the solver kernels inside one LM iteration, from a rocprofv3 kernel trace.
    rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -- python3 tools/time_solver.py C2 8 --hybrid
    python tools/studies/level_times.py "/tmp/lt/**/*kernel_trace.csv"
Prints, for the median iteration (delimited by k_accept), every launch in order: kernel, grid (workgroups), duration, gap to
the previous launch's end; then per kernel name the list of per-launch medians over all iterations (one entry per level)."""
import csv
import glob
import sys
from collections import defaultdict

import numpy as np

path = glob.glob(sys.argv[1], recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")
its, cur = [], []
for r in rows:
    if not name(r).startswith("k_"):
        continue
    cur.append(r)
    if name(r).startswith("k_accept"):
        its.append(cur)
        cur = []
lens = [len(i) for i in its]
L = int(np.median(lens))
its = [i for i in its if len(i) == L][2:]
print(f"{len(its)} iterations of {L} launches")
dur = np.array([[(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in it] for it in its])
gap = np.array([[0.0] + [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(it, it[1:])] for it in its])
med, gmed = np.median(dur, axis=0), np.median(gap, axis=0)
per = defaultdict(list)
for k, r in enumerate(its[0]):
    wg = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"]))
    print(f"  {k:3d} {name(r)[:22]:22s} wg {wg:7d}  {med[k]:8.1f} us   gap {gmed[k]:6.1f}")
    per[name(r)].append((wg, med[k]))
print(f"sum of durations {med.sum():.1f} us, sum of gaps {gmed.sum():.1f} us")
for n, v in per.items():
    print(f"{n[:24]:24s} " + "  ".join(f"{wg}:{d:.0f}" for wg, d in v) + f"   = {sum(d for _, d in v):.0f} us")
