"""Study: how good is the elimination ordering of the multifrontal solver, and what would a better one buy?

    python tools/studies/nd_order_study.py [C2] [seeds=4]          (CPU only)

Builds tools/studies/nd/nd_stats.cpp + python-super_amd/csrc/slm_nd_host.hip with g++ (the symbolic analysis is host
code), hands it the coupling graph of synthetic frames (surfel KNN tuples + node KNN, as slm_bind_frame does) and prints
the plan's cost figures: factorisation FLOPs (64-padded, as the tile kernels execute them, and exact), fronts, levels,
pivot-column chain of the top fronts.  Environment switches of the analysis (SLM_ND_*) select ordering variants."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "python-super_amd"))
from super_amd import synth  # noqa: E402


def build():
    out = os.path.join(ROOT, "tools", "studies", "nd", "_nd_stats.so")
    src = [os.path.join(ROOT, "tools", "studies", "nd", "nd_stats.cpp"), os.path.join(ROOT, "python-super_amd", "csrc", "slm_nd_host.hip")]
    if not os.path.exists(out) or any(os.path.getmtime(s) > os.path.getmtime(out) for s in src + [os.path.join(ROOT, "python-super_amd", "csrc", "slm_nd.h")]):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "python-super_amd", "csrc"),
                               "-x", "c++", *src, "-o", out])
    return C.CDLL(out)


def graph_of(sc):
    J = sc.J
    idx = np.asarray(sc.sf_knn_idx, dtype=np.int64)
    tup = np.unique(np.sort(idx, axis=1), axis=0)
    keys = set()
    for a in range(tup.shape[1]):
        for b in range(tup.shape[1]):
            hi, lo = np.maximum(tup[:, a], tup[:, b]), np.minimum(tup[:, a], tup[:, b])
            keys.update((hi * J + lo).tolist())
    pairs = np.array(sorted(keys), dtype=np.uint32)
    return (np.ascontiguousarray(sc.ed_points, dtype=np.float32), np.ascontiguousarray(sc.ed_knn_idx, dtype=np.int32), pairs)


def stats(lib, J, pts, knn, pairs):
    out = (C.c_double * 8)()
    fr = np.zeros((4096, 4), dtype=np.int32)
    kp = np.zeros(4096, dtype=np.int32)
    n = lib.nd_stats(J, knn.shape[1], pts.ctypes.data_as(C.c_void_p), knn.ctypes.data_as(C.c_void_p), pairs.ctypes.data_as(C.c_void_p),
                     len(pairs), out, fr.ctypes.data_as(C.c_void_p), 4096, kp.ctypes.data_as(C.c_void_p))
    assert n > 0
    stats.kp = kp[:n]
    return list(out), fr[:n]


def main():
    wl = sys.argv[1] if len(sys.argv) > 1 else "C2"
    seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    lib = build()
    tot = np.zeros(8)
    for s in range(seeds):
        sc = synth.make_scene(seed=s, **synth.WORKLOADS[wl])
        pts, knn, pairs = graph_of(sc)
        o, fr = stats(lib, sc.J, pts, knn, pairs)
        tot += np.array(o)
        if s == 0:
            top = fr[fr[:, 0] <= 2]
            print("top fronts (depth, pivots nv, boundary nb): " + ", ".join(f"d{d}:{nv}/{nb}" for d, nv, nb, _ in top))
            npt = [(7 * nv + 63) // 64 for d, nv, nb, _ in fr]
            chain = {}
            for (d, nv, nb, _), k in zip(fr, npt):
                chain[d] = max(chain.get(d, 0), k)
            print("longest pivot-column chain per depth:", [chain[d] for d in sorted(chain)], "sum", sum(chain.values()))
        print(f"seed {s}: padded {o[0] / 1e9:.3f} GF  exact {o[1] / 1e9:.3f} GF  fronts {int(o[2])} levels {int(o[3])}  modelled critical {o[4]:.0f} us  "
              f"storage {o[5] / 1e6:.0f} MB  tasks {int(o[6])} items {int(o[7])}")
    tot /= seeds
    print(f"mean   : padded {tot[0] / 1e9:.3f} GF  exact {tot[1] / 1e9:.3f} GF  critical {tot[4]:.0f} us  storage {tot[5] / 1e6:.0f} MB  tasks {tot[6]:.0f}")


if __name__ == "__main__":
    main()


def per_depth(fr):
    """padded / exact FLOPs per depth of the fronts fr (depth, nv, nb, parent)"""
    r64 = lambda x: (x + 63) // 64 * 64
    rows = {}
    for d, nv, nb, _ in fr:
        n1, n2 = 7.0 * nv, 7.0 * nb
        p1, p2 = float(r64(7 * nv)), float(r64(7 * nb))
        e = n1 ** 3 / 3 + n1 * n1 * n2 + n1 * n2 * n2
        p = p1 ** 3 / 3 + p1 * p1 * p2 + p1 * p2 * p2
        a = rows.setdefault(int(d), [0, 0.0, 0.0, [], []])
        a[0] += 1; a[1] += p; a[2] += e; a[3].append(int(nv)); a[4].append(int(nb))
    for d in sorted(rows):
        n, p, e, nvs, nbs = rows[d]
        print(f"  depth {d}: {n:3d} fronts  padded {p / 1e9:6.3f} GF  exact {e / 1e9:6.3f} GF  nv {min(nvs)}..{max(nvs)} (mean {np.mean(nvs):.1f})  nb {min(nbs)}..{max(nbs)} (mean {np.mean(nbs):.1f})")
