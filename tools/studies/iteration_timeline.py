"""Diagnostic: start / end of every kernel of ONE LM iteration (all streams) from a rocprofv3 kernel trace.
    rocprofv3 --kernel-trace --output-format csv -d /tmp/lt -- python3 tools/time_solver.py C2 8 --hybrid
    python tools/studies/iteration_timeline.py "/tmp/lt/**/*kernel_trace.csv" [iteration=30]"""
import csv
import glob
import sys

path = glob.glob(sys.argv[1], recursive=True)[0]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")
idx = [i for i, r in enumerate(rows) if name(r).startswith("k_iter_begin_nd")]
a, b = idx[k], idx[k + 1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = 0.0
for r in rows[a:b]:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    q = r.get("Queue_Id", "?")
    print(f"{name(r)[:20]:20s} q{q:>3s} start {s:8.1f} end {e:8.1f} dur {e - s:7.1f}  {'(overlaps the previous)' if s < prev_end - 0.5 else ''}")
    prev_end = max(prev_end, e)
