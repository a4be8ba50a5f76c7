#!/bin/bash
# Same-box A/B of the headline bench over one environment switch: tools/ab_env.sh VAR A_VALUE B_VALUE [rounds] [bench args]
V=$1; A=$2; B=$3; R=${4:-3}; shift 4
for i in $(seq $R); do
  for x in "$A" "$B"; do
    env $V=$x timeout -k 10 200 python bench.py --no-cpu-baseline --no-kernel-events --steps 20 "$@" 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$V=$x', round(d['ms_per_step'],3), 'ms/step', round(d['value'],1))" || exit 1
  done
done
