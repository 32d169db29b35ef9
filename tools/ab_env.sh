#!/bin/bash
# Headline step under several environment settings, alternating, on one box:
#   bash tools/ab_env.sh REPS "A3D_X=1 A3D_Y=2" "A3D_X=0" ...
reps=$1; shift
for i in $(seq $reps); do
  for cfg in "$@"; do
    env $cfg python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-fine --no-dp-rank --also "" 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print('$cfg', d['ms_per_step'], d['value'])"
  done
done
