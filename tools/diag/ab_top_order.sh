#!/bin/bash
# Diagnostic: the headline with the hybrid's top list ordered by its own model (default) against the whole-tree order filtered
# by depth (SLM_DAG_TOP_LEGACY=1), alternated.
run() {
  v=$(env "$@" python3 bench.py --no-cpu-baseline --no-latency-b1 --steps 12 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f it/s  solve %.3f ms  step med %.2f ms  worst status %d' % (d['value'], d['roofline']['avg_phase_ms'], d['step_ms']['median'], d['worst_iter_status_all_ranks']))")
  echo "$* -> $v"
}
for i in 1 2 3; do run X=0; run SLM_DAG_TOP_LEGACY=1; done
for wl in C1 C4; do
  for e in X=0 SLM_DAG_TOP_LEGACY=1; do
    v=$(env $e python3 bench.py --workload $wl --no-cpu-baseline --no-latency-b1 --steps 8 --warmup 2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f it/s  solve %.3f ms' % (d['value'], d['roofline']['avg_phase_ms']))")
    echo "$wl $e -> $v"
  done
done
