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
    assert C.sizeof(SlmGfConfig) == 10 * 4 + 8 * 8
    assert C.sizeof(SlmGfSemantic) == 2 * 4 + 5 * 8
    assert C.sizeof(SlmFrame) == 7 * 4 + 4 * 4 + 4 + 9 * 8 + 2 * 4   # 4 bytes padding before pointers; state_f64 + pad


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
    # the stages either side of the solve refuse as well
    h = C.c_void_p()
    assert lib.slm_fuse_create(48, 64, 1000, C.byref(h)) != 0 and b"no HIP device" in lib.slm_last_error()
    assert lib.slm_depth_create(48, 64, C.byref(h)) != 0 and b"no HIP device" in lib.slm_last_error()
    from super_amd.graph_encoder import DirectDeformGraph
    from types import SimpleNamespace
    with pytest.raises(SuperLMError):
        DirectDeformGraph(SimpleNamespace(method="super"))


def test_abi_handshake(lib):
    """A caller built against another revision of the header gets an error, not out-of-bounds accesses."""
    from super_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "super_lm.h")).read()
    assert int(re.search(r"#define SLM_ABI_VERSION (\d+)", hdr).group(1)) == _lib.SLM_ABI_VERSION == lib.slm_abi_version()
    assert int(re.search(r"#define SLM_PLAN_INFO_DOUBLES (\d+)", hdr).group(1)) == _lib.PLAN_INFO_DOUBLES
    sizes = [C.sizeof(t) for t in (_lib.SlmConfig, _lib.SlmFrame, _lib.SlmGfConfig, _lib.SlmGfFrame, _lib.SlmIterRecord)]
    assert lib.slm_abi_check(_lib.SLM_ABI_VERSION, *sizes) == 0
    assert lib.slm_abi_check(_lib.SLM_ABI_VERSION - 1, *sizes) != 0 and b"slm_abi_check" in lib.slm_last_error()
    r01_frame = sizes[1] - 8             # slm_frame before state_f64 / pad were added
    assert lib.slm_abi_check(_lib.SLM_ABI_VERSION, sizes[0], r01_frame, *sizes[2:]) != 0
    # out-of-range paths are rejected, not silently mapped to a default (checked before the device test)
    cfg = _lib.SlmConfig(num_iterations=10, phase_test=1, use_data=1, use_arap=1, use_rot=1, max_frames=1,
                         solver_path=7, w_data=1.0, w_arap=10.0, w_rot=1.0, u0=10.0, v=7.5, minimal_loss0=1e10)
    out = C.c_void_p()
    assert lib.slm_create(C.byref(cfg), C.byref(out)) == 1 and b"solver_path" in lib.slm_last_error()


def test_argument_validation(lib):
    assert lib.slm_create(None, None) != 0
    assert lib.slm_destroy(None) == 0
    assert lib.slm_knn(10, 0, 4, 0, None, None, None, None, None) != 0
    assert b"slm_knn" in lib.slm_last_error()


def test_every_struct_matches_the_header_as_compiled_by_gcc(tmp_path):
    """sizeof() of every struct of include/super_lm.h, compiled as plain C, equals the ctypes mirror
    (a maintainer's cgo / ctypes binding sees exactly these layouts)."""
    import subprocess
    from super_amd import _lib
    pairs = {"slm_config": _lib.SlmConfig, "slm_frame": _lib.SlmFrame, "slm_iter_record": _lib.SlmIterRecord,
             "slm_gf_config": _lib.SlmGfConfig, "slm_gf_frame": _lib.SlmGfFrame, "slm_gf_semantic": _lib.SlmGfSemantic,
             "slm_depth_config": _lib.SlmDepthConfig, "slm_depth_inputs": _lib.SlmDepthInputs,
             "slm_depth_outputs": _lib.SlmDepthOutputs, "slm_fuse_config": _lib.SlmFuseConfig,
             "slm_surfel_model": _lib.SlmSurfelModel, "slm_new_frame": _lib.SlmNewFrame,
             "slm_fuse_semantic": _lib.SlmFuseSemantic,
             "slm_graph_outputs": _lib.SlmGraphOutputs}
    src = tmp_path / "sizes.c"
    body = "\n".join(f'  printf("{n} %zu\\n", sizeof({n}));' for n in pairs)
    src.write_text('#include <stdio.h>\n#include "super_lm.h"\nint main(void) {\n' + body + "\n  return 0;\n}\n")
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                   check=True)
    out = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for name, ct in pairs.items():
        assert int(out[name]) == C.sizeof(ct), (name, out[name], C.sizeof(ct))
