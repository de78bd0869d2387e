#!/bin/bash
# Diagnostic: k_data_gram ablations (stamps build, SLM_DBG bits: 1 skip record/slab flush,
# 2 skip MFMA, 4 skip surfel evaluation) under rocprofv3 kernel stats.  Run on the GPU box:
#   gpurun -- 'bash tools/ablate_gram.sh'   ->  gpurun_out/abl_<bits>/**/..._kernel_stats.csv
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export SLM_LIB=$R/python-super_amd/lib/libsuper_lm_stamps.so
for d in 0 1 2 4 3 7; do
  export SLM_DBG=$d
  timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abl_$d -- python3 $R/tools/time_assemble.py 30 > $R/gpurun_out/abl_$d.log 2>&1 < /dev/null
  echo "DBG=$d rc=$?"
  find $R/gpurun_out/abl_$d -name '*kernel_trace.csv' -delete 2>/dev/null
done
