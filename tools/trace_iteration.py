"""Diagnostic: print the kernel timeline of one LM iteration from a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile
    python tools/trace_iteration.py gpurun_out/trace
"""
import csv
import glob
import sys

f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_data_gram" in r["Kernel_Name"]]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]["Start_Timestamp"])
agg = {}
for r in rows[a:b]:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    s = int(r["Start_Timestamp"]) - t0
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg.setdefault(n, [0, 0])
    agg[n][0] += 1
    agg[n][1] += d
    if len(sys.argv) > 2:
        print(f"{s/1e3:9.1f} us {d/1e3:8.1f} us  {n[:28]:28s} grid {r['Grid_Size_X']},{r['Grid_Size_Y']},{r['Grid_Size_Z']}")
tot = int(rows[b]["Start_Timestamp"]) - t0
busy = sum(v[1] for v in agg.values())
print(f"iteration {tot/1e3:.1f} us, kernels busy {busy/1e3:.1f} us, launches {b-a}")
for n, (c, d) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n[:32]:32s} x{c:4d} {d/1e3:9.1f} us {100*d/tot:5.1f}%")
