#!/bin/bash
# Diagnostic: the headline (C2, 8 frames per launch) under the environment settings given as arguments, alternated with the default.
#   bash tools/diag/ab_env.sh "SLM_X=1" "SLM_X=2" ...
run() {
  v=$(env "$@" python3 bench.py --no-cpu-baseline --no-latency-b1 --steps 12 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f it/s  solve %.3f ms  step med %.2f ms  worst status %d' % (d['value'], d['roofline']['avg_phase_ms'], d['step_ms']['median'], d['worst_iter_status_all_ranks']))")
  echo "$* -> $v"
}
for rep in 1 2; do
  run X=0
  for e in "$@"; do run $e; done
done
