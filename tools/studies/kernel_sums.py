"""Diagnostic: per-kernel time per LM iteration from a rocprofv3 --kernel-trace --stats run.
    rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 tools/time_solver.py C2 8 --hybrid
    python tools/studies/kernel_sums.py "/tmp/ks/**/*kernel_stats.csv" [top] [--sq profiles/rNN_pmc_sq_summary.csv] [--md]
One line per kernel: calls, us per iteration (iterations = launches of k_accept), average us per launch; with --sq the
MFMA-busy, waiting and VALU-active shares of the SQ-counter summary (profiles/make_sq_summary.py) beside them; --md prints
the table as markdown (what DESIGN.md section 4 quotes: regenerate it from the round's profiles, do not edit it by hand)."""
import csv
import glob
import sys

path = glob.glob(sys.argv[1], recursive=True)[0] if "*" in sys.argv[1] else sys.argv[1]
args = [a for a in sys.argv[2:] if not a.startswith("--")]
sq_path = sys.argv[sys.argv.index("--sq") + 1] if "--sq" in sys.argv else None
if sq_path in args:
    args.remove(sq_path)
top = int(args[0]) if args else 16
sq = {}
if sq_path:
    for r in csv.DictReader(open(sq_path)):
        sq[r["kernel"]] = r
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
md = "--md" in sys.argv
if md:
    print("| kernel | launches / iteration | us / iteration | us / launch | MFMA-busy | waiting | VALU-active |")
    print("|---|---|---|---|---|---|---|")
for us, name, calls, avg in sorted(out, reverse=True)[:top]:
    q = sq.get(name.split("<")[0], {})
    pct = lambda k: (f"{100 * float(q[k]):.1f} %" if q.get(k) not in (None, "") else "--")
    if md:
        print(f"| `{name.split('<')[0]}` | {calls / n:.1f} | {us:.0f} | {avg:.1f} | {pct('mfma_busy_share')} | {pct('wait_share')} | {pct('valu_active_share')} |")
    else:
        extra = f"  mfma {pct('mfma_busy_share'):>7s} wait {pct('wait_share'):>7s} valu {pct('valu_active_share'):>7s}" if sq else ""
        print(f"  {name[:28]:28s} {calls:6d} calls {us:8.1f} us/iter {avg:8.1f} us avg{extra}")
