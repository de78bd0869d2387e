#!/bin/bash
# Diagnostic (GPU box): per-kernel us per LM iteration for a library variant.
#   bash tools/studies/kernel_sums.sh [libname.so] [workload] [frames]
LIB=${1:-libsuper_lm.so}; WL=${2:-C2}; B=${3:-8}
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=$(mktemp -d /tmp/ks.XXXX)
cd /tmp && export TMPDIR=/tmp
SLM_LIB=$LIB rocprofv3 --kernel-trace --stats --output-format csv -d $D -- python3 $R/tools/time_solver.py $WL $B --hybrid > $D/log.txt 2>&1
grep "solver_path" $D/log.txt
python3 $R/tools/studies/kernel_sums.py "$D/**/*kernel_stats.csv" ${4:-14}
rm -rf $D
