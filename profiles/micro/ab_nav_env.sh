#!/bin/bash
# same-box A/B of a navigator-iteration switch: bash profiles/micro/ab_nav_env.sh VAR valueA valueB [rounds] [extra bench_nav flags]
V=$1; A=$2; B=$3; N=${4:-3}; shift 4
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O
for i in $(seq 1 $N); do
  for x in $A $B; do
    env $V=$x python bench_nav.py --steps 10 --warmup 10 --no-cpu-baseline --no-host-loop --no-profile "$@" > $O/ab_nav_tmp.json 2>/dev/null || exit 1
    python -c "import json;d=json.load(open('$O/ab_nav_tmp.json'));print('$V=$x', d['ms_per_step'], d['value'])" | tee -a $O/ab_nav_$V.txt
  done
done
