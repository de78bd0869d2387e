"""Diagnostic: per-kernel time per LM iteration from a rocprofv3 --kernel-trace --stats run.
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 tools/time_solver.py C2 8 --hybrid
    python tools/studies/kernel_sums.py "/tmp/ks/**/*kernel_stats.csv" [top]
One line per kernel: calls, us per iteration (iterations = launches of k_accept), average us per launch."""
import csv
import glob
import sys

path = glob.glob(sys.argv[1], recursive=True)[0] if "*" in sys.argv[1] else sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 16
rows = list(csv.DictReader(open(path)))
n = [int(r["Calls"]) for r in rows if r["Name"].startswith("k_accept")][0]
tot = 0.0
out = []
for r in rows:
    name = r["Name"].split("(")[0].replace("void ", "")
    if not name.startswith("k_"):
        continue
    us = float(r["TotalDurationNs"]) / n / 1e3
    tot += us
    out.append((us, name, int(r["Calls"]), float(r["AverageNs"]) / 1e3))
print(f"iterations {n}; k_* kernels {tot:.1f} us per iteration")
for us, name, calls, avg in sorted(out, reverse=True)[:top]:
    print(f"  {name[:28]:28s} {calls:6d} calls {us:8.1f} us/iter {avg:8.1f} us avg")
