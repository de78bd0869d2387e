#!/usr/bin/env python
"""HBM traffic per launch of the kernels that tools/profile_rows.py runs (GraphFit, the Semantic-SuPer step, depth
preprocessing, fusion, the ED graph, the K-generic LM path), from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
collected in SEPARATE runs; units and gfx950 corrections as profiles/make_traffic.py: KiB, FETCH x 2, WRITE x 1).

    python profiles/make_rows_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> --out profiles/r06_rows_pmc_traffic.json

Dispatches are grouped by (kernel name with its template arguments, grid size): GraphFit at one and at eight frames per
launch are different groups of the same kernel.  `graphfit_c2` sums a whole slm_gf_run (bench.graphfit_timing: 10 Adam
iterations) at the smallest (one frame) and the largest (eight frames) grid of k_gf_data<4, false>."""
import argparse
import collections
import csv
import json
import os
import sys


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    depth, out = 0, []
    for ch in n:             # cut the argument list, keep the template arguments
        if ch == "(" and depth == 0:
            break
        depth += ch == "<"
        depth -= ch == ">"
        out.append(ch)
    return "".join(out).strip()


def groups(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[(short(r["Kernel_Name"]), int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("--out", required=True)
    ap.add_argument("--iterations", type=int, default=10, help="optimiser iterations per slm_gf_run of the profiled script")
    a = ap.parse_args()
    f, w = groups(a.fetch_csv), groups(a.write_csv)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out = {"lib_sha16": bench.lib_sha16(), "correction": "FETCH_SIZE x2 (gfx950 wide-read under-count), WRITE_SIZE x1; KiB -> bytes",
           "kernels": []}
    per = {}
    for k in sorted(set(f) & set(w)):
        if not (k[0].startswith("k_") or k[0].startswith("kb_")):
            continue
        b = (2.0 * f[k][0] + w[k][0]) * 1024.0
        per[k] = b
        out["kernels"].append({"kernel": k[0], "grid_size": k[1], "launches_sampled": f[k][1], "FETCH_SIZE_KiB": f[k][0],
                               "WRITE_SIZE_KiB": w[k][0], "traffic_bytes_per_launch": b})
    # one slm_gf_run at C2 (plain options): k_gf_zero once, then per iteration k_gf_data<4, false> (the node terms ride on it) + k_gf_step
    gd = sorted(g for (n, g) in per if n == "k_gf_data<4, false>")
    if gd:
        gf = {}
        for tag, pick in (("b1", min), ("b8", max)):
            tot = a.iterations * per[("k_gf_data<4, false>", pick(gd))]
            for name in ("k_gf_step", "k_gf_zero", "k_gf_advance", "k_gf_reg"):
                gs = sorted(g for (n, g) in per if n == name)
                if gs:
                    tot += per[(name, pick(gs))] * (a.iterations if name in ("k_gf_step", "k_gf_reg") else 1)
            gf[tag] = {"traffic_bytes_per_run": tot, "iterations": a.iterations}
        out["graphfit_c2"] = gf
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out.get("graphfit_c2"), indent=1))


if __name__ == "__main__":
    main()
