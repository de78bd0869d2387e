#!/usr/bin/env python
"""Copy what `profiles/collect.sh <tag>` left under gpurun_out/ into profiles/ (tracked):
    python profiles/install.py r02
bench lines, kernel-stat summaries (8 frames and 1 frame per launch), the two HBM PMC passes trimmed to the columns
used, the per-kernel traffic summary (re-made here so that its tag is the hash of the sources in this tree:
run it BEFORE touching csrc/ again) and the SQ-counter summary."""
import csv
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
G = os.path.join(ROOT, "gpurun_out")
P = os.path.join(ROOT, "profiles")
KEEP = ["Dispatch_Id", "Grid_Size", "Kernel_Name", "Workgroup_Size", "LDS_Block_Size", "VGPR_Count", "Accum_VGPR_Count",
        "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"]


def one(pattern):
    m = glob.glob(os.path.join(G, pattern), recursive=True)
    if not m:
        raise SystemExit("missing " + pattern)
    return max(m, key=os.path.getmtime)      # gpurun merges into gpurun_out/: older collections may still be there


shutil.copy(os.path.join(G, f"{tag}_bench.json"), os.path.join(P, f"{tag}_C2_b8_bench.json"))
shutil.copy(os.path.join(G, f"{tag}_bench_b1.json"), os.path.join(P, f"{tag}_C2_b1_bench.json"))
shutil.copy(one(f"{tag}_stats/**/*kernel_stats.csv"), os.path.join(P, f"{tag}_C2_b8_kernel_stats.csv"))
shutil.copy(one(f"{tag}_stats_b1/**/*kernel_stats.csv"), os.path.join(P, f"{tag}_C2_b1_kernel_stats.csv"))
for wl in ("C1", "C4"):
    src = os.path.join(G, f"{tag}_bench_{wl}.json")
    if os.path.exists(src) and os.path.getsize(src) > 0:
        shutil.copy(src, os.path.join(P, f"{tag}_{wl}_b8_bench.json"))
trimmed = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    src = one(f"{tag}_pmc_{c}/**/*counter_collection.csv")
    dst = os.path.join(P, f"{tag}_C2_b8_pmc_{c}.csv")
    with open(src) as f, open(dst, "w", newline="") as g:
        w = csv.DictWriter(g, KEEP)
        w.writeheader()
        for r in csv.DictReader(f):
            w.writerow({k: r[k] for k in KEEP})
    trimmed[c] = dst
subprocess.check_call([sys.executable, os.path.join(P, "make_traffic.py"), trimmed["FETCH_SIZE"], trimmed["WRITE_SIZE"],
                       "--workload", "C2", "--frames-per-gpu", "8", "--out", os.path.join(P, f"{tag}_pmc_traffic.json")])
subprocess.check_call([sys.executable, os.path.join(P, "make_sq_summary.py")] +
                      sorted(glob.glob(os.path.join(G, f"{tag}_pmc_sq_*/"))) + ["--out", os.path.join(P, f"{tag}_pmc_sq_summary.csv")])
rows = glob.glob(os.path.join(G, f"{tag}_stats_rows/**/*kernel_stats.csv"), recursive=True)
if rows:     # GraphFit / semantic GraphFit / depth / fusion / graph / K = 6 kernels (tools/profile_rows.py), our kernels only
    with open(max(rows, key=os.path.getmtime)) as f, open(os.path.join(P, f"{tag}_rows_kernel_stats.csv"), "w", newline="") as g:
        rd = csv.DictReader(f)
        w = csv.DictWriter(g, rd.fieldnames)
        w.writeheader()
        for r in rd:
            nm = r["Name"].replace("void ", "")
            if "k_" in nm or "kb_" in nm or "rocprim" in nm:       # (some kernels live in anonymous namespaces)
                w.writerow(r)
    if os.path.exists(os.path.join(G, f"{tag}_rows.json")):
        lines = [ln for ln in open(os.path.join(G, f"{tag}_rows.json")) if ln.startswith("{")]
        if lines:
            open(os.path.join(P, f"{tag}_rows_timing.json"), "w").write(lines[-1])
rf = glob.glob(os.path.join(G, f"{tag}_rows_pmc_FETCH_SIZE/**/*counter_collection.csv"), recursive=True)
rw = glob.glob(os.path.join(G, f"{tag}_rows_pmc_WRITE_SIZE/**/*counter_collection.csv"), recursive=True)
if rf and rw:   # HBM bytes per launch of the same rows (re-made here: tagged with the hash of the sources in this tree)
    subprocess.check_call([sys.executable, os.path.join(P, "make_rows_traffic.py"), max(rf, key=os.path.getmtime), max(rw, key=os.path.getmtime),
                           "--out", os.path.join(P, f"{tag}_rows_pmc_traffic.json")])
rsq = sorted(glob.glob(os.path.join(G, f"{tag}_rows_sq_*/")))
if rsq:
    subprocess.check_call([sys.executable, os.path.join(P, "make_sq_summary.py")] + rsq +
                          ["--keep-templates", "--out", os.path.join(P, f"{tag}_rows_sq_summary.csv")])
avail = os.path.join(G, f"{tag}_mfma_counters_available.txt")
if os.path.exists(avail):
    shutil.copy(avail, os.path.join(P, f"{tag}_mfma_counters_available.txt"))
print("installed", sorted(os.path.basename(p) for p in glob.glob(os.path.join(P, tag + "_*"))))
