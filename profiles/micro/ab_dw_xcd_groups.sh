#!/bin/bash
# round 6: the weight-gradient launch with a split's tiles on XCDs of its own (csrc/gemm.hip dw_xcd_groups) and 1 / 2 / 4 / 8 K-splits, against the old
# (tile, split) order with 3 splits -- step time and the launch's own duration (bench.py's instrumented pass), same box, interleaved.
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; rm -f $O/r06_ab_dw_xcd.txt
run() { # label, env...
  local label=$1; shift
  env "$@" timeout -k 10 300 python bench.py --no-parity --no-secondary --no-cpu-baseline > $O/ab_dw_tmp.json 2>> $O/r06_ab_dw_xcd.err || return 1
  python -c "import json;d=json.load(open('$O/ab_dw_tmp.json'));w=d['roofline']['detail']['weight_gradient_launch'];print('$label', d['ms_per_step'], d['steady']['ms_per_step'], d['steady']['ms_per_step_by_task'], 'dW launch us', w['avg_us'], 'alg MB', round(w['algorithmic_bytes_per_launch']/1e6,1))" | tee -a $O/r06_ab_dw_xcd.txt
}
for rep in 1 2; do
run "old(order,3 splits)" MAGIC_DW_XCD_GROUPS=0 MAGIC_SPLITK_POW2=0 || exit 1
run "groups,pow2-up     " MAGIC_DW_XCD_GROUPS=1 MAGIC_SPLITK_POW2=up || exit 1
run "groups,pow2-down   " MAGIC_DW_XCD_GROUPS=1 MAGIC_SPLITK_POW2=down || exit 1
run "old order,pow2-up  " MAGIC_DW_XCD_GROUPS=0 MAGIC_SPLITK_POW2=up || exit 1
done
