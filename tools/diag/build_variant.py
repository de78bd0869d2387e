"""Diagnostic: build lib/libsuper_lm_<tag>.so = the normal objects with ONE source recompiled under extra -D flags.
    python tools/diag/build_variant.py <tag> <source.hip> -DNAME=VALUE ...
Run a script against it with SLM_LIB=libsuper_lm_<tag>.so (super_amd/_lib.py)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "python-super_amd"))
from super_amd import build as b
tag, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
b.build()
objs = [os.path.join(b.OBJ_DIR, os.path.basename(s)[:-4] + ".o") for s in b.sources()]
vobj = os.path.join(b.OBJ_DIR, f"{src[:-4]}_{tag}.o")
subprocess.run([b.HIPCC] + b.FLAGS + flags + ["-c", os.path.join(b.CSRC, src), "-o", vobj], check=True)
objs = [vobj if os.path.basename(o) == src[:-4] + ".o" else o for o in objs]
out = os.path.join(b.LIB_DIR, f"libsuper_lm_{tag}.so")
subprocess.run([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, check=True)
print(out)
