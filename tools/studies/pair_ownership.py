"""Study (CPU, NumPy): how many coupled node pairs of the data term are owned by ONE workgroup of k_data_gram?
(VERDICT r01 item 4 proposed writing single-workgroup pairs straight into the fronts from the kernel's epilogue.)
Restates the layout of slm_prep.hip: canonical 4-tuples sorted, each tuple's surfels padded to a multiple of 4
positions, 256 positions per workgroup; a (workgroup, pair) record exists for every pair of every tuple the workgroup
touches.      python tools/studies/pair_ownership.py [C1|C2|C4]
C2: 11055 tuples, 853 workgroups, 15471 pairs, 42779 records (= the plan's own counts in the bench line);
    pairs with 1 / 2 / 3 / 4 / 5+ records: 3890 / 3370 / 4049 / 2199 / 1963 -- single-workgroup pairs are 25 % of the
    pairs and 9 % of the records."""
import os
import sys
from collections import defaultdict

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "python-super_amd"))
from super_amd import synth  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
sc = synth.make_scene(seed=0, **synth.WORKLOADS[wl])
knn = np.sort(np.asarray(sc.sf_knn_idx).astype(np.int64), axis=1)
key = ((knn[:, 0] * 65536 + knn[:, 1]) * 65536 + knn[:, 2]) * 65536 + knn[:, 3]
order = np.argsort(key, kind="stable")
uniq, start, cnt = np.unique(key[order], return_index=True, return_counts=True)
pc = (cnt + 3) & ~3
pstart = np.concatenate([[0], np.cumsum(pc)[:-1]])
npos = int(pc.sum())
nodes = knn[order][start]
wgs = defaultdict(set)
for t in range(len(uniq)):
    w0, w1 = pstart[t] // 256, (pstart[t] + pc[t] - 1) // 256
    n = nodes[t]
    for a in range(4):
        for b in range(a + 1):
            wgs[(int(n[a]), int(n[b]))].update(range(w0, w1 + 1))
c = np.array([len(v) for v in wgs.values()])
print(f"{wl}: tuples {len(uniq)}, positions {npos}, workgroups {(npos + 255) // 256}, pairs {len(c)}, records {c.sum()}")
print(f"   pairs owned by one workgroup: {(c == 1).sum()} = {100 * (c == 1).mean():.1f} % of the pairs, "
      f"{100 * (c == 1).sum() / c.sum():.1f} % of the records")
print("   records per pair histogram (1, 2, ...):", np.bincount(c)[1:12].tolist())
