#!/bin/bash
# SQ / TCC counters of the generation kernels (development; through gpurun from the repo root):
#   bash tools/gen_counters.sh TAG   -> gpurun_out/TAG_gen_sq.{json,txt}, gpurun_out/TAG_gen_traffic.{json,txt}
TAG=${1:-gen}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/${TAG}_gen_pmc_SQ -o p -- python $ROOT/bench.py --mode generate --gen-sentences 1000 --no-cpu-baseline > /dev/null 2>&1 || exit 1
python $ROOT/tools/pmc_summary.py $OUT/${TAG}_gen_sq.json SQ=$OUT/${TAG}_gen_pmc_SQ > $OUT/${TAG}_gen_sq.txt
rm -rf $OUT/${TAG}_gen_pmc_SQ
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 240 rocprofv3 --pmc $c --output-format csv -d $OUT/${TAG}_gen_pmc_$c -o p -- python $ROOT/bench.py --mode generate --gen-sentences 1000 --no-cpu-baseline > /dev/null 2>&1 || exit 1
done
python $ROOT/tools/pmc_summary.py $OUT/${TAG}_gen_traffic.json FETCH_SIZE=$OUT/${TAG}_gen_pmc_FETCH_SIZE WRITE_SIZE=$OUT/${TAG}_gen_pmc_WRITE_SIZE > $OUT/${TAG}_gen_traffic.txt
rm -rf $OUT/${TAG}_gen_pmc_FETCH_SIZE $OUT/${TAG}_gen_pmc_WRITE_SIZE
cat $OUT/${TAG}_gen_sq.txt $OUT/${TAG}_gen_traffic.txt
