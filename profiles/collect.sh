#!/bin/bash
# Collect the round's judged profiles on the GPU box (run through gpurun from the repo root):
#   rm -rf gpurun_out/<tag>_*; gpurun --timeout 2100 -- 'bash profiles/collect.sh <tag>'
# Writes under gpurun_out/<tag>_*; `python profiles/install.py <tag>` then copies the summaries into profiles/.
# rocprofv3 rules on this pool: the program itself after `--`; --pmc passes separate from --stats.
# Order: the HBM counter passes first, so that the bench line written afterwards can quote `roofline_data.traffic`
# from a summary of exactly this build (bench.py only uses a summary whose source hash matches the tree).
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
# (SLM_FUSE_BEGIN=0 for the counter passes only: the iteration's zeroing as a launch of its own, so that its bytes are not
#  booked on k_data_gram, whose launch carries it in the default build -- round 6; same sources, same hash)
export SLM_FUSE_BEGIN=0
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "k_" --output-format csv -d $R/gpurun_out/${TAG}_pmc_$c -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > $R/gpurun_out/${TAG}_pmc_$c.log 2>&1 < /dev/null
  find $R/gpurun_out/${TAG}_pmc_$c -name '*kernel_trace.csv' -delete
done
F=$(find $R/gpurun_out/${TAG}_pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)
W=$(find $R/gpurun_out/${TAG}_pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python3 $R/profiles/make_traffic.py $F $W --workload C2 --frames-per-gpu 8 --out $R/gpurun_out/${TAG}_pmc_traffic.json > /dev/null
cp $R/gpurun_out/${TAG}_pmc_traffic.json $R/profiles/${TAG}_pmc_traffic.json      # (on the box; install.py makes the tracked copy)
unset SLM_FUSE_BEGIN
# the same two passes over the rows the LM bench does not run (tools/profile_rows.py: GraphFit, the semantic step, depth, fusion,
# graph, K = 6) -- before the bench, whose roofline_graphfit.traffic quotes the summary
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --kernel-trace --kernel-include-regex "k_" --output-format csv -d $R/gpurun_out/${TAG}_rows_pmc_$c -- python3 $R/tools/profile_rows.py > $R/gpurun_out/${TAG}_rows_pmc_$c.log 2>&1 < /dev/null
  find $R/gpurun_out/${TAG}_rows_pmc_$c -name '*kernel_trace.csv' -delete
done
RF=$(find $R/gpurun_out/${TAG}_rows_pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)
RW=$(find $R/gpurun_out/${TAG}_rows_pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python3 $R/profiles/make_rows_traffic.py $RF $RW --out $R/gpurun_out/${TAG}_rows_pmc_traffic.json > /dev/null
cp $R/gpurun_out/${TAG}_rows_pmc_traffic.json $R/profiles/${TAG}_rows_pmc_traffic.json
timeout 400 python3 $R/bench.py > $R/gpurun_out/${TAG}_bench.json 2> $R/gpurun_out/${TAG}_bench.err < /dev/null
timeout 300 python3 $R/bench.py --frames-per-gpu 1 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_b1.json 2> $R/gpurun_out/${TAG}_bench_b1.err < /dev/null
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --no-cpu-baseline --no-latency-b1 > $R/gpurun_out/${TAG}_stats.log 2>&1 < /dev/null
find $R/gpurun_out/${TAG}_stats -name '*kernel_trace.csv' -delete
# SQ counters of the same command (wave lifetime, waiting share, VALU share): two small passes
# (+ the MFMA pass: MFMA-busy cycles against the kernel's own cycles, and the f64 MFMA op count where the counter exists)
rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*" | sort -u > $R/gpurun_out/${TAG}_mfma_counters_available.txt
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA"; do
  n=$(echo $grp | cut -d' ' -f1)
  timeout 400 rocprofv3 --pmc $grp --kernel-trace --kernel-include-regex "k_" --output-format csv -d $R/gpurun_out/${TAG}_pmc_sq_$n -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-profile > $R/gpurun_out/${TAG}_pmc_sq_$n.log 2>&1 < /dev/null
  find $R/gpurun_out/${TAG}_pmc_sq_$n -name '*kernel_trace.csv' -delete
done
python3 $R/profiles/make_sq_summary.py $R/gpurun_out/${TAG}_pmc_sq_*/ --out $R/gpurun_out/${TAG}_pmc_sq_summary.csv > /dev/null 2>&1
# one frame per launch (the drop-in case): kernel stats of the task-graph solver
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats_b1 -- python3 $R/bench.py --frames-per-gpu 1 --no-cpu-baseline --no-latency-b1 > $R/gpurun_out/${TAG}_stats_b1.log 2>&1 < /dev/null
find $R/gpurun_out/${TAG}_stats_b1 -name '*kernel_trace.csv' -delete
# the other synthetic workloads (bench lines only)
timeout 400 python3 $R/bench.py --workload C1 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_C1.json 2> $R/gpurun_out/${TAG}_bench_C1.err < /dev/null
timeout 600 python3 $R/bench.py --workload C4 --no-cpu-baseline > $R/gpurun_out/${TAG}_bench_C4.json 2> $R/gpurun_out/${TAG}_bench_C4.err < /dev/null
# the rows the LM bench's trace does not contain: GraphFit at C2 (1 and 8 frames per launch), the Semantic-SuPer
# GraphFit step at C4 (configs[4]), depth preprocessing, fusion + swap, ED-graph construction, the K-generic LM path at K = 6
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats_rows -- python3 $R/tools/profile_rows.py > $R/gpurun_out/${TAG}_rows.json 2> $R/gpurun_out/${TAG}_rows.err < /dev/null
find $R/gpurun_out/${TAG}_stats_rows -name '*kernel_trace.csv' -delete
# ... and their SQ counters (the HBM passes of the same script ran first, above)
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"; do
  n=$(echo $grp | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $grp --kernel-trace --kernel-include-regex "k_" --output-format csv -d $R/gpurun_out/${TAG}_rows_sq_$n -- python3 $R/tools/profile_rows.py > $R/gpurun_out/${TAG}_rows_sq_$n.log 2>&1 < /dev/null
  find $R/gpurun_out/${TAG}_rows_sq_$n -name '*kernel_trace.csv' -delete
done
ls $R/gpurun_out/${TAG}_stats/* $R/gpurun_out/${TAG}_pmc_FETCH_SIZE/* < /dev/null
