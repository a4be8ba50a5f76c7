#!/bin/bash
# Regenerates the round's evidence on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh TAG   ->  gpurun_out/TAG_{bench.json,kernel_stats.csv,pmc_traffic.json,pmc_traffic.txt,gaps.txt}
# The program under rocprofv3 is python itself (no env/bash hop); each PMC pass is its own run.  The kernel trace is
# of the default (graph-replayed) bench; the counter passes launch the same kernels eagerly (--no-graph).
set -u
TAG=${1:-r01_x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 python $ROOT/bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -o t -- python $ROOT/bench.py --no-cpu-baseline --driver-sentences 0 > /dev/null 2>&1 || exit 1
cp $OUT/${TAG}_trace/t_kernel_stats.csv $OUT/${TAG}_kernel_stats.csv
python $ROOT/tools/trace_gaps.py $OUT/${TAG}_trace/t_kernel_trace.csv --last 8 > $OUT/${TAG}_gaps.txt 2>&1
rm -rf $OUT/${TAG}_trace
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d $OUT/${TAG}_pmc_$c -o p -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-graph --driver-sentences 0 > /dev/null 2>&1 || exit 1
done
python $ROOT/tools/pmc_summary.py $OUT/${TAG}_pmc_traffic.json FETCH_SIZE=$OUT/${TAG}_pmc_FETCH_SIZE WRITE_SIZE=$OUT/${TAG}_pmc_WRITE_SIZE > $OUT/${TAG}_pmc_traffic.txt
rm -rf $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE
# SQ counters per kernel (its own pass): matrix-pipe utilisation and wait share, the recurrences included
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/${TAG}_pmc_SQ -o p -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --no-graph --driver-sentences 0 > /dev/null 2>&1 || exit 1
python $ROOT/tools/pmc_summary.py $OUT/${TAG}_sq_counters.json SQ=$OUT/${TAG}_pmc_SQ > $OUT/${TAG}_sq_counters.txt
rm -rf $OUT/${TAG}_pmc_SQ
cut -c1-300 $OUT/${TAG}_bench.json
head -12 $OUT/${TAG}_gaps.txt
