"""Diagnostic: idle gaps and long kernels in a rocprofv3 --kernel-trace csv.
    python tools/studies/trace_gaps.py <kernel_trace.csv> [gap_us=1000] [long_us=3000]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
gap_us = float(sys.argv[2]) if len(sys.argv) > 2 else 1000.0
long_us = float(sys.argv[3]) if len(sys.argv) > 3 else 3000.0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
prev_end, prev = t0, None
for i, r in enumerate(rows):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if (s - prev_end) / 1e3 > gap_us:
        print(f"t={(s - t0) / 1e6:9.2f} ms  GAP {(s - prev_end) / 1e3:9.1f} us  after {name(prev) if prev else '-'}  before {name(r)}")
    if (e - s) / 1e3 > long_us:
        print(f"t={(s - t0) / 1e6:9.2f} ms  LONG {(e - s) / 1e3:8.1f} us  {name(r)}  grid {r.get('Grid_Size_X', '?')}")
    if e > prev_end:
        prev_end, prev = e, r
print("kernels", len(rows), "span ms", (prev_end - t0) / 1e6)
