#!/bin/bash
# Generation (BASELINE configs[4]) on the GPU box: bench line, per-kernel stats and a window of the kernel
# timeline.  Run through gpurun from the repo root:  bash tools/gen_profile.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out/genprof
timeout -k 10 200 python $R/bench.py --mode generate > $R/gpurun_out/genprof/bench.json 2> $R/gpurun_out/genprof/bench.err || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/genprof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/genprof -- python $R/bench.py --mode generate > /tmp/genprof.log 2>&1 || exit 1
cp /tmp/genprof/*/*kernel_stats.csv $R/gpurun_out/genprof/kernel_stats.csv
python $R/tools/trace_window.py /tmp/genprof/*/*kernel_trace.csv --at 0.1 --n 12 > $R/gpurun_out/genprof/window.txt
# the same at eval_bs 1000 (the tiled GEMM + LSE kernels instead of the decode-batch kernels)
rm -rf /tmp/genprof1000
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/genprof1000 -- python $R/bench.py --mode generate --gen-batch 1000 --no-cpu-baseline > $R/gpurun_out/genprof/bench_eval_bs_1000.json 2> /tmp/genprof1000.log || exit 1
cp /tmp/genprof1000/*/*kernel_stats.csv $R/gpurun_out/genprof/kernel_stats_eval_bs_1000.csv
cat $R/gpurun_out/genprof/bench.json $R/gpurun_out/genprof/window.txt
