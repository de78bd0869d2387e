#!/bin/bash
# Diagnostic: the headline (C2, 8 frames per launch) under the solver's environment knobs, one short bench run each.
# usage (through gpurun): bash tools/diag/knob_sweep.sh > gpurun_out/knobs.log
run() {
  v=$(env "$@" python3 bench.py --no-cpu-baseline --no-latency-b1 --steps 12 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f it/s  solve %.3f ms  step med %.2f ms' % (d['value'], d['roofline']['avg_phase_ms'], d['step_ms']['median']))")
  echo "$* -> $v"
}
run X=0
run SLM_ND_LEAF=14
run SLM_ND_LEAF=24
run SLM_ND_LEAF=32
run SLM_DAG_TOP_FRONTS=2
run SLM_DAG_TOP_FRONTS=8
run SLM_COMPACT_NPT=3
run SLM_COMPACT_NPT=5
run SLM_COMPACT_MIN=32
run SLM_COMPACT_MIN=128
run SLM_BACK_FUSE_NPT=1
run SLM_BACK_FUSE_NPT=3
run SLM_DAG_DEFER_BOUNDARY=1
run SLM_DAG_DEFER_BOUNDARY=3
run X=1
