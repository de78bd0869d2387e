"""CPU-side checks of the C-ABI library: it builds, loads, and exports every symbol that
include/super_lm.h declares.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from super_amd import build, _lib
    build.build()
    return _lib.load()


def test_header_symbols_are_exported(lib):
    from super_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "super_lm.h")).read()
    declared = set(re.findall(r"^(?:int|const char\*)\s+(slm_\w+)\s*\(", hdr, flags=re.M))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), f"{name} not exported by libsuper_lm.so"


def test_struct_layouts_match_header(lib):
    from super_amd._lib import SlmConfig, SlmFrame, SlmIterRecord
    assert C.sizeof(SlmConfig) == 8 * 4 + 6 * 8
    assert C.sizeof(SlmIterRecord) == 2 * 8 + 4 * 4
    from super_amd._lib import SlmGfConfig, SlmGfSemantic
    assert C.sizeof(SlmGfConfig) == 10 * 4 + 7 * 8
    assert C.sizeof(SlmGfSemantic) == 2 * 4 + 5 * 8
    assert C.sizeof(SlmFrame) == 7 * 4 + 4 * 4 + 4 + 9 * 8   # 4 bytes padding before pointers


def test_no_device_fails_loudly(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from super_amd._lib import SlmConfig, SuperLMError, check
    cfg = SlmConfig(num_iterations=10, phase_test=1, use_data=1, use_arap=1, use_rot=1, max_frames=1,
                    w_data=1.0, w_arap=10.0, w_rot=1.0, u0=10.0, v=7.5, minimal_loss0=1e10)
    out = C.c_void_p()
    rc = lib.slm_create(C.byref(cfg), C.byref(out))
    assert rc != 0                      # SLM_ERR_NO_DEVICE, never a silent CPU path
    with pytest.raises(SuperLMError):
        check(rc, "slm_create")
    from super_amd.LM import LM_Solver
    from oracle.lm_oracle import default_opt
    with pytest.raises(SuperLMError):
        LM_Solver(default_opt())


def test_argument_validation(lib):
    assert lib.slm_create(None, None) != 0
    assert lib.slm_destroy(None) == 0
    assert lib.slm_knn(10, 0, 4, 0, None, None, None, None, None) != 0
    assert b"slm_knn" in lib.slm_last_error()
