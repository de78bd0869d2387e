"""Diagnostic: timeline of ONE LM iteration from a rocprofv3 --kernel-trace csv.
    rocprofv3 --kernel-trace -d gpurun_out/tl -o tl --output-format csv -- python3 tools/time_solver.py C2 8
    python tools/studies/trace_timeline.py gpurun_out/tl/*/tl_kernel_trace.csv [nth-from-last data_gram]
Prints, from one k_data_gram to the next, every dispatch: start offset (us), duration, gap before it, name."""
import csv
import glob
import sys

path = glob.glob(sys.argv[1], recursive=True)[0] if "*" in sys.argv[1] else sys.argv[1]
nth = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void k_data_gram") or r["Kernel_Name"].startswith("k_data_gram")]
a, b = idx[-nth - 1], idx[-nth]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
busy = {}
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]
    g = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {(s - prev_end) / 1e3:6.1f}  grid {g:>8}  {name}")
    prev_end = max(prev_end, e)
    busy[name] = busy.get(name, 0) + (e - s)
print(f"iteration span {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us; busy by kernel:")
for k, v in sorted(busy.items(), key=lambda kv: -kv[1]):
    print(f"   {v / 1e3:8.1f} us  {k}")
