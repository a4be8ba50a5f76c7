#!/bin/bash
# Same-box A/B of the generation bench: _ab/ (a git worktree of an earlier commit with its own built library) against
# the working tree, alternating.  usage (through gpurun): tools/ab_gen.sh [rounds]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq ${1:-2}); do
  for t in _ab .; do
    (cd $ROOT/$t && timeout -k 10 200 python bench.py --mode generate --no-cpu-baseline 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$t', round(d['value'],1), 'sent/s  argmax', round(r['avg_launch_us'],2), 'us', r['kernel'])") || exit 1
  done
done
