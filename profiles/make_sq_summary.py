#!/usr/bin/env python
"""Per-kernel means of the SQ counters collected by profiles/collect.sh (separate rocprofv3 --pmc passes):

    python profiles/make_sq_summary.py gpurun_out/r02_pmc_sq_*/ --out profiles/r02_pmc_sq_summary.csv

Columns: dispatches, then per counter the mean per dispatch; derived: cycles per wave (SQ_WAVE_CYCLES / SQ_WAVES),
waiting share (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES) and VALU-active share (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES)
when the counters are present.  Wave-cycle counters are in units of 4 cycles on gfx9 (quad-cycle granularity);
the shares are ratios of like units.  mfma_busy_share = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)
when both were collected in the same pass (the MFMA utilisation north_star asks for)."""
import argparse
import collections
import csv
import glob
import os


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dirs", nargs="+")
    ap.add_argument("--out", required=True)
    ap.add_argument("--keep-templates", action="store_true", help="group by the kernel name WITH its template arguments")
    a = ap.parse_args()
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in a.dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(path)):
                name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
                if a.keep_templates:       # cut the argument list only
                    depth, cut = 0, len(name)
                    for i, ch in enumerate(name):
                        if ch == "(" and depth == 0:
                            cut = i
                            break
                        depth += (ch == "<") - (ch == ">")
                    name = name[:cut].strip()
                else:
                    name = name.split("(")[0].split("<")[0]
                agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    counters = sorted({c for k in agg.values() for c in k})
    with open(a.out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "dispatches"] + counters + ["cycles_per_wave", "wait_share", "valu_active_share", "mfma_busy_share"])
        for k in sorted(agg, key=lambda k: -sum(agg[k].get("SQ_WAVE_CYCLES", [0]))):
            m = {c: (sum(v) / len(v)) for c, v in agg[k].items()}
            n = max(len(v) for v in agg[k].values())
            wc, wv = m.get("SQ_WAVE_CYCLES"), m.get("SQ_WAVES")
            row = [k, n] + [f"{m[c]:.1f}" if c in m else "" for c in counters]
            row.append(f"{wc / wv:.1f}" if wc and wv else "")
            row.append(f"{m['SQ_WAIT_INST_ANY'] / wc:.3f}" if wc and "SQ_WAIT_INST_ANY" in m else "")
            row.append(f"{m['SQ_ACTIVE_INST_VALU'] / wc:.3f}" if wc and "SQ_ACTIVE_INST_VALU" in m else "")
            # MFMA utilisation: MFMA-busy cycles summed over the SIMDs / (kernel cycles x 1024 SIMDs); kernel cycles =
            # GRBM_GUI_ACTIVE / 8 (rocprofv3 reports the sum over the 8 XCDs, MI355X_MICROARCH.md "DVFS give-back")
            ga, mb = m.get("GRBM_GUI_ACTIVE"), m.get("SQ_VALU_MFMA_BUSY_CYCLES")
            row.append(f"{mb / (ga / 8.0 * 1024.0):.4f}" if ga and mb is not None else "")
            w.writerow(row)
    print(a.out)


if __name__ == "__main__":
    main()
