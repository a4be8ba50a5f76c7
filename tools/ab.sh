#!/bin/bash
# Same-box A/B of the headline bench: the tree in _ab/ (a git worktree of an earlier commit with its own built
# library) against the working tree, alternating, N rounds.  usage (through gpurun): tools/ab.sh [rounds] [bench args]
R=${1:-3}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for i in $(seq $R); do
  for t in _ab .; do
    (cd $ROOT/$t && timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-events --steps 20 "$@" 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$t', round(d['ms_per_step'],3), 'ms/step', round(d['value'],1))") || exit 1
  done
done
