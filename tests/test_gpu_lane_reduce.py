"""The VALU-only lane reduction of the task-graph solver's back substitution (csrc/slm_lane.h: v_permlane32_swap,
v_permlane16_swap, DPP) against the portable __shfl_xor butterfly: same sums in the same lanes.  Built from
tools/micro/col_reduce_mb.hip on the GPU box (hipcc is part of the image); the solver's own parity tests cover it
end to end, this one pins the lane mapping."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_col_reduce16_matches_shuffle_butterfly(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "col_reduce_mb")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-I", os.path.join(ROOT, "python-super_amd", "csrc"),
                    "-o", exe, os.path.join(ROOT, "tools", "micro", "col_reduce_mb.hip")],
                   check=True, capture_output=True, timeout=300)
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=120).stdout
    lines = out.splitlines()
    assert "result lanes" in lines[0] and ": ok" in lines[0], out          # col_reduce16 vs the shuffle butterfly
    assert "col_reduce8" in lines[1] and ": ok" in lines[1], out           # col_reduce8 vs plain half-wave sums
