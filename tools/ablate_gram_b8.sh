#!/bin/bash
# Diagnostic (GPU box): k_data_gram ablations at B = 8 (stamps build, SLM_DBG bits: 1 skip the record / slab flush,
# 2 skip the MFMAs, 4 skip the row computation; SLM_NO_REUSE=1: the pass runs in every iteration although the ablated
# runs reject their steps) -- per-kernel us per LM iteration of tools/time_solver.py under rocprofv3.
#   gpurun -- 'bash tools/ablate_gram_b8.sh'
R=${GRAFT_REPO_ROOT:-$(pwd)}
python3 -c "
import sys; sys.path.insert(0, '$R/python-super_amd')
from super_amd import build; print(build.build(stamps=True))"
for d in 0 1 2 4 3 7; do
  echo "== SLM_DBG=$d"
  SLM_NO_REUSE=1 SLM_DBG=$d bash $R/tools/studies/kernel_sums.sh libsuper_lm_stamps.so C2 8 40 2>&1 | grep -E "k_data_gram|k_data_eval|per iteration"
done
