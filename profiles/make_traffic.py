#!/usr/bin/env python
"""Summarise rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs,
as MI355X_MICROARCH.md prescribes) into per-kernel HBM traffic per launch.

    python profiles/make_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> \
        --workload C2 --frames-per-gpu 8 --out profiles/r01_pmc_traffic.json

Units and gfx950 corrections (MI355X_MICROARCH.md, "HBM"): the counters are in KiB;
FETCH_SIZE reports half of the bytes of a wide coalesced read, so it is doubled;
WRITE_SIZE is exact.  traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch.
"""
import argparse
import collections
import csv
import json


# kernels of each phase of an LM iteration on the nested-dissection path (csrc/slm_api.hip slm_run)
PHASE_KERNELS = {
    "zero": ["k_iter_begin_nd", "k_zero_f22"],
    "data_grad": ["k_data_gram"],
    "reg_grad": ["k_front_assemble", "k_reg_grad_nd", "k_front_load_rhs"],
    "solve": ["k_fL11", "k_fL21", "k_fpanel", "k_fpotrf", "k_ftrsm", "k_ftrail", "k_fschur", "k_fpull", "k_fdag", "k_dag_reset",
              "k_dag_check", "k_fback_prep", "k_fbacksub", "k_make_trial"],
    "data_loss": ["k_data_loss", "k_data_eval"],   # (round 4: k_data_eval is the loss pass on the tuple-sorted path)
    "accept": ["k_reg_loss", "k_accept"],
}


def means(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        agg[name].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_csv")
    ap.add_argument("write_csv")
    ap.add_argument("--workload", default="C2")
    ap.add_argument("--frames-per-gpu", type=int, default=8)
    ap.add_argument("--out", required=True)
    ap.add_argument("--lib", default=None, help="(ignored: the tag is the hash of the library SOURCES, bench.lib_sha16) "
                                                "bench.py can tell a summary of an older build from a current one)")
    a = ap.parse_args()
    f, w = means(a.fetch_csv), means(a.write_csv)
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    sha = bench.lib_sha16()      # hash of csrc/ + include/: the build that was profiled
    out = {"workload": a.workload, "frames_per_gpu": a.frames_per_gpu, "lib_sha16": sha, "kernels": {}}
    for k in sorted(set(f) & set(w)):
        fetch_kib, n = f[k]
        write_kib, _ = w[k]
        out["kernels"][k] = {
            "launches_sampled": n, "FETCH_SIZE_KiB": fetch_kib, "WRITE_SIZE_KiB": write_kib,
            "traffic_bytes_all_launches": (2.0 * fetch_kib + write_kib) * 1024.0 * n,
            "traffic_bytes_per_launch": (2.0 * fetch_kib + write_kib) * 1024.0,
            "correction": "FETCH_SIZE x2 (gfx950 wide-read under-count), WRITE_SIZE x1"}
    # per LM iteration and phase (the phases of include/super_lm.h SLM_PH_*): every launch of the phase's kernels in the
    # sampled run / the LM iterations of that run (k_accept runs once per iteration)
    n_iter = out["kernels"].get("k_accept", {}).get("launches_sampled", 0)
    if n_iter:
        out["iterations_sampled"] = n_iter
        out["phases"] = {}
        for ph, names in PHASE_KERNELS.items():
            ks = {k: out["kernels"][k]["traffic_bytes_all_launches"] / n_iter for k in names if k in out["kernels"]}
            out["phases"][ph] = {"traffic_bytes_per_iteration": sum(ks.values()), "kernels": ks}
        out["traffic_bytes_per_iteration_all_phases"] = sum(p["traffic_bytes_per_iteration"] for p in out["phases"].values())
    json.dump(out, open(a.out, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
