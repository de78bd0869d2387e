"""The binned preparation (round 4: bins per node, one LDS sort per bin, 19 launches) against the rocPRIM pipeline of
rounds 1-3 (four device-wide sorts, ~88 launches), which stays in the library as the path for degenerate inputs
(``SLM_PREP_LEGACY=1`` forces it).  Both implement the model-side half of ``loss_term.prepare`` (reference
``super/loss.py:178-197,212-220``: the scatter structure of the Jacobian) and must give the SAME plan, array for array --
then every floating-point summation order downstream is unchanged too.  On an MI355X (-m gpu): fresh child processes,
because the switch is read once per process."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r'''
import ctypes as C, hashlib, json, sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1] + "/python-super_amd"); sys.path.insert(0, sys.argv[1])
from super_amd import _lib, synth
from super_amd.engine import DeviceFrame, Engine
dev = torch.device("cuda", 0)
cases = json.loads(sys.argv[2])
out = {}
for name, kw in cases.items():
    sc = synth.make_scene(**kw)
    for f64 in (False, True):
        eng = Engine(dev, data_path=0, num_iterations=3)
        fr = DeviceFrame.from_scene(sc, dev, state_f64=f64)
        eng.bind(0, fr)
        eng.bind(0, fr)                      # the second bind takes the hinted path
        info = eng.plan_info(0)
        digest = {}
        for what in range(13):
            n = C.c_int64(0)
            _lib.check(eng.lib.slm_debug_read_plan(eng.h, 0, what, None, 0, C.byref(n), eng.stream), "size")
            buf = (C.c_char * max(n.value, 1))()
            _lib.check(eng.lib.slm_debug_read_plan(eng.h, 0, what, buf, n.value, C.byref(n), eng.stream), "read")
            digest[str(what)] = [n.value, hashlib.sha256(bytes(buf[:n.value])).hexdigest()]
        eng.run(1)
        beta = eng.beta(0).cpu().numpy()
        out[f"{name}/{'f64' if f64 else 'f32'}"] = {"info": {k: v for k, v in info.items() if k != "solver"}, "plan": digest,
                                                    "beta_sha": hashlib.sha256(beta.tobytes()).hexdigest(),
                                                    "loss": [r["loss"] for r in eng.records(0)]}
        eng.close()
print("RESULT " + json.dumps(out))
'''

CASES = {
    "tiny": dict(N=3000, J=48, H=60, W=80, seed=7, src_border=5, tgt_border=3),
    "small": dict(N=12000, J=108, H=120, W=160, seed=3, src_border=6, tgt_border=4),
    "c1": dict(N=50000, J=512, H=480, W=640, seed=1),
}


def _run(legacy):
    env = dict(os.environ)
    env.pop("SLM_PREP_LEGACY", None)
    if legacy:
        env["SLM_PREP_LEGACY"] = "1"
    p = subprocess.run([sys.executable, "-c", _CHILD, ROOT, json.dumps(CASES)], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


def test_binned_preparation_gives_the_rocprim_plan_array_for_array():
    new, old = _run(False), _run(True)
    assert new.keys() == old.keys()
    for key in new:
        assert new[key]["info"] == old[key]["info"], key
        for what, (n_bytes, sha) in new[key]["plan"].items():
            assert [n_bytes, sha] == old[key]["plan"][what], (key, "plan array", what, n_bytes, old[key]["plan"][what][0])
        assert new[key]["info"]["tuples"] > 0 and new[key]["info"]["merged_records"] > 0


def test_a_bin_that_does_not_fit_falls_back_to_the_rocprim_pipeline():
    """60 000 surfels on 12 nodes: thousands of surfels per smallest node -- the binned preparation hands the slot to
    the rocPRIM pipeline, for this and the following binds; results as before."""
    import torch
    from super_amd import synth
    from super_amd.engine import DeviceFrame, Engine
    dev = torch.device("cuda", 0)
    sc = synth.make_scene(N=60000, J=12, H=240, W=320, seed=5, src_border=6, tgt_border=4)
    eng = Engine(dev, num_iterations=2)
    fr = DeviceFrame.from_scene(sc, dev)
    eng.bind(0, fr)
    eng.run(1)
    b0 = eng.beta(0).cpu().numpy()
    eng.bind(0, fr)
    eng.run(1)
    b1 = eng.beta(0).cpu().numpy()
    assert [r["status"] for r in eng.records(0)] == [0, 0]
    assert np.abs(b0 - b1).max() < 1e-9 and np.isfinite(b0).all()
    small = synth.make_scene(N=3000, J=48, H=60, W=80, seed=7, src_border=5, tgt_border=3)
    eng.bind(0, DeviceFrame.from_scene(small, dev))            # the slot stays on the rocPRIM pipeline: still correct
    eng.run(1)
    assert [r["status"] for r in eng.records(0)] == [0, 0]
    eng.close()
