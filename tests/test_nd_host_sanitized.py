"""The host-side symbolic analysis of the multifrontal solver (python-super_amd/csrc/slm_nd_host.hip: plain C++, runs
inside slm_bind_frame) under AddressSanitizer + UBSan on degenerate coupling graphs: one node, random graphs with
self references and duplicates, duplicate points, disconnected islands, invalid (-1) node-KNN entries, a dense graph.
CPU only (GPU sanitizers are not available on the pool); the harness is tools/studies/nd/nd_stats.cpp."""
import os
import shutil
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = textwrap.dedent('''
    import ctypes as C, sys
    import numpy as np
    lib = C.CDLL(sys.argv[1])
    rng = np.random.default_rng(0)
    def run(J, K, pts, knn, pairs):
        out = (C.c_double * 8)(); fr = np.zeros((8192, 4), np.int32)
        pts = np.ascontiguousarray(pts, np.float32); knn = np.ascontiguousarray(knn, np.int32)
        pairs = np.ascontiguousarray(pairs, np.uint32).reshape(-1)
        n = lib.nd_stats(J, K, pts.ctypes.data_as(C.c_void_p), knn.ctypes.data_as(C.c_void_p), pairs.ctypes.data_as(C.c_void_p),
                         len(pairs), out, fr.ctypes.data_as(C.c_void_p), 8192, None)
        assert n >= 1, (J, n)
        assert sum(int(f[1]) for f in fr[:n]) == J, "every node is a pivot of exactly one front"
        rc = lib.nd_check_orders(J, K, pts.ctypes.data_as(C.c_void_p), knn.ctypes.data_as(C.c_void_p), pairs.ctypes.data_as(C.c_void_p), len(pairs))
        assert rc == 0, ("task list is not a valid ticket order (a task precedes something it waits for)", J, rc)
        return n
    run(1, 4, np.zeros((1, 3)), np.zeros((1, 4), np.int32), np.array([0]))
    run(2, 1, rng.normal(size=(2, 3)), np.array([[1], [0]]), np.array([0, 2, 3]))
    for J in (5, 19, 40, 200, 700):
        K = 4
        a = rng.integers(0, J, size=6 * J); b = rng.integers(0, J, size=6 * J)
        run(J, K, rng.normal(size=(J, 3)), rng.integers(0, J, size=(J, K)), np.unique(np.maximum(a, b).astype(np.int64) * J + np.minimum(a, b)))
        pts = rng.uniform(size=(J, 3)); pts[: J // 4] = pts[0]; pts[J // 2:] += 100      # duplicates + an island
        if J <= 400:
            d = ((pts[:, None] - pts[None]) ** 2).sum(-1)
            knn = np.argsort(d, axis=1, kind="stable")[:, 1:K + 1]
            pr = sorted({max(i, int(j)) * J + min(i, int(j)) for i in range(J) for j in knn[i]})
            assert run(J, K, pts, knn, np.array(pr)) >= 1
    run(30, 4, rng.normal(size=(30, 3)), np.full((30, 4), -1), np.zeros(0))
    run(60, 4, rng.normal(size=(60, 3)), rng.integers(0, 60, size=(60, 4)), np.array([a * 60 + b for a in range(60) for b in range(a + 1)]))
    print("sanitized analysis ok")
''')


def test_symbolic_analysis_is_clean_under_asan_and_ubsan(tmp_path):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("no g++")
    so = str(tmp_path / "nd_asan.so")
    csrc = os.path.join(ROOT, "python-super_amd", "csrc")
    subprocess.check_call([gxx, "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
                           "-std=c++17", "-shared", "-fPIC", "-I", csrc, "-x", "c++",
                           os.path.join(ROOT, "tools", "studies", "nd", "nd_stats.cpp"), os.path.join(csrc, "slm_nd_host.hip"), "-o", so])
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    if not os.path.isabs(asan):
        pytest.skip("libasan not found")
    drv = tmp_path / "drv.py"
    drv.write_text(DRIVER)
    env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, str(drv), so], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "sanitized analysis ok" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
