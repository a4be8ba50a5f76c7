#!/bin/bash
# The bench's other configurations on the GPU box (run through gpurun from the repo root):
#   tools/variants_round.sh TAG -> gpurun_out/TAG_{adv,configs0,disentangle,syntaxvae,varlen,stress}.json
TAG=${1:-r01_x}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
run() { n=$1; shift; timeout -k 10 400 python $R/bench.py --no-cpu-baseline "$@" > $O/${TAG}_$n.json 2> $O/${TAG}_$n.err || { echo "$n failed"; tail -5 $O/${TAG}_$n.err; exit 1; }; python -c "
import json; d=json.loads(open('$O/${TAG}_$n.json').read().strip().splitlines()[-1]); print('$n', round(d['value'],1), round(d['ms_per_step'],3))"; }
run adv --adv-train && run configs0 --model VanillaVAE --batch 32 --vocab 10000 && run disentangle --model DisentangleVAE && \
run syntaxvae --model SyntaxVAE && run varlen --variable-lengths && run stress --stress --batch 1024 --steps 5 --warmup 3
