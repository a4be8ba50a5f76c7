#!/bin/bash
# Same-box A/B of the generation bench over one environment switch: tools/ab_gen_env.sh VAR A_VALUE B_VALUE [rounds]
V=$1; A=$2; B=$3; R=${4:-3}
for i in $(seq $R); do
  for x in "$A" "$B"; do
    env $V=$x timeout -k 10 200 python bench.py --mode generate 2>/dev/null \
      | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline',{}); print('$V=$x', round(d['value'],1), 'sentences/s; arg-max', round(r.get('avg_launch_us',0),2), 'us', r.get('padded_rows',{}).get('rows'))" || exit 1
  done
done
