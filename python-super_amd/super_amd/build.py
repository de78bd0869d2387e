"""Build libsuper_lm.so (gfx950 only) in-tree with hipcc.

    python -m super_amd.build            # from python-super_amd/
    python python-super_amd/super_amd/build.py

The shared library lands in ``python-super_amd/lib/`` (git-ignored, but it travels
with the gpurun snapshot).  hipcc cross-compiles for gfx950 without a GPU.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))   # python-super_amd/
REPO_ROOT = os.path.dirname(PKG_ROOT)
CSRC = os.path.join(PKG_ROOT, "csrc")
LIB_DIR = os.path.join(PKG_ROOT, "lib")
OBJ_DIR = os.path.join(PKG_ROOT, "build")
LIB_PATH = os.path.join(LIB_DIR, "libsuper_lm.so")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=fast-honor-pragmas",
         "-I" + os.path.join(REPO_ROOT, "include"), "-I" + CSRC]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, stamps: bool = False) -> str:
    """``stamps=True`` builds the DIAGNOSTIC variant (in-kernel s_memtime stamps, printed by
    slm_get_records); never ship or time that build."""
    obj_dir, lib_path = OBJ_DIR, LIB_PATH          # (locals: a stamps build must not redirect later normal builds)
    if stamps:
        obj_dir = os.path.join(PKG_ROOT, "build", "stamps")
        lib_path = os.path.join(LIB_DIR, "libsuper_lm_stamps.so")
    os.makedirs(LIB_DIR, exist_ok=True)
    os.makedirs(obj_dir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(REPO_ROOT, "include", "super_lm.h"))
    srcs = sources()
    objs = [os.path.join(obj_dir, os.path.basename(s)[:-4] + ".o") for s in srcs]

    def compile_one(pair):
        src, obj = pair
        if not force and not _stale(obj, [src] + headers):
            return None
        cmd = [HIPCC] + FLAGS + (["-DSLM_STAMPS"] if stamps else []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        return r.stderr

    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        logs = list(ex.map(compile_one, zip(srcs, objs)))
    if force or _stale(lib_path, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        for lg in logs:
            if lg:
                print(lg)
    return lib_path


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, stamps="--stamps" in sys.argv))
