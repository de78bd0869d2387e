"""Diagnostic: what runs BETWEEN two LM runs of the bench step (bind, update, copies) from a rocprofv3 --kernel-trace csv
of `python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-latency-b1 --no-profile`.
    python tools/studies/trace_step.py <kernel_trace.csv>
Prints the span from the last k_accept of one step to the first k_data_gram of the next: per-kernel busy time, the union
busy time (streams overlap), idle time."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("void ", "")[:44]
acc = [i for i, r in enumerate(rows) if name(r).startswith("k_accept")]
# steps: groups of 10 k_accept; take the gap after the 10th accept of the second-to-last step
groups = [acc[i:i + 10] for i in range(0, len(acc), 10)]
g = groups[-2]
a = g[-1]
b = next(i for i in range(a, len(rows)) if name(rows[i]).startswith("k_data_gram"))
t0 = int(rows[a]["End_Timestamp"])
t1 = int(rows[b]["Start_Timestamp"])
busy = {}
iv = []
for r in rows[a + 1:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    busy.setdefault(name(r), [0, 0])
    busy[name(r)][0] += e - s
    busy[name(r)][1] += 1
    iv.append((s, e))
iv.sort()
union = 0
cs, ce = None, None
for s, e in iv:
    if cs is None:
        cs, ce = s, e
    elif s <= ce:
        ce = max(ce, e)
    else:
        union += ce - cs
        cs, ce = s, e
if cs is not None:
    union += ce - cs
print(f"between two LM runs: {(t1 - t0) / 1e3:.1f} us, GPU busy (union over streams) {union / 1e3:.1f} us, launches {len(iv)}")
for k, (v, n) in sorted(busy.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"  {v / 1e3:9.1f} us  x{n:4d}  {k}")
lm0 = int(rows[groups[-2][0]]["Start_Timestamp"])
print(f"LM run itself (first to last k_accept of the step): {(int(rows[g[-1]]['End_Timestamp']) - lm0) / 1e3:.1f} us")
