#!/bin/bash
# Study (1-GPU box): N bench ranks sharing cuda:0 (BENCH_SHARE_GPU=1, gloo exchange) -- the host-CPU pattern of an N-GPU job
# under the container's CPU quota: CFS throttling counters before / after, with the spinning and the polled drain.
#   bash tools/studies/multirank_cpu.sh [ranks=4] [steps=10]
# Result (round 3): inconclusive for the drain -- under gloo the host-staged all-gather (`local.cpu()`) spins by itself, 3.2-3.5
# CPUs per rank either way, 3-4 throttled periods per run (the start-up).  Side observation: four processes of 8 frames on
# ONE GPU reach 3 260-3 400 it/s in total (one process of 8 frames: 2 740): their latency- and throughput-bound phases overlap.
N=${1:-4}; K=${2:-10}
thr() { grep -E "nr_throttled|throttled_usec" /sys/fs/cgroup/cpu.stat | tr '\n' ' '; }
for v in 1 0 1 0; do
  a=$(thr)
  SLM_SPIN_WAIT=$v BENCH_SHARE_GPU=1 BENCH_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
    --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 400)) bench.py --gpus $N --steps $K --warmup 2 --no-cpu-baseline --no-latency-b1 2>/dev/null | tail -1 | \
    python -c "import sys,json; d=json.loads(sys.stdin.read()); print('spin', $v, 'value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), 'cpu/rank', d['host']['cpu_cores_busy_per_rank'])"
  echo "   cpu.stat before: $a"
  echo "   cpu.stat after : $(thr)"
done
