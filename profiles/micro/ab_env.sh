#!/bin/bash
# same-box A/B of one environment switch: bash profiles/micro/ab_env.sh NAME VALUE_A VALUE_B [rounds]
cd $GRAFT_REPO_ROOT
F="--steps 60 --warmup 10 --no-cpu-baseline --no-parity --no-secondary --no-profile"
show='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], j["ms_per_step"], j["steady"]["ms_per_step"])'
for i in $(seq 1 ${4:-3}); do
  for v in $2 $3; do
    env $1=$v python bench.py $F 2>/dev/null | python -c "$show" "$1=$v"
  done
done
