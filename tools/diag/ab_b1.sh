#!/bin/bash
# Diagnostic: one frame per launch (the task-graph solver) under the environment settings given as arguments, alternated with the default.
run() {
  v=$(env "$@" python3 bench.py --frames-per-gpu 1 --no-cpu-baseline --no-latency-b1 --steps 20 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.0f it/s  solve %.3f ms  worst status %d' % (d['value'], d['roofline']['avg_phase_ms'], d['worst_iter_status_all_ranks']))")
  echo "b1 $* -> $v"
}
for rep in 1 2; do
  run X=0
  for e in "$@"; do run $e; done
done
