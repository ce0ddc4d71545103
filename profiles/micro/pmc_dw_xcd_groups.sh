#!/bin/bash
# round 6: L2-miss traffic of the weight-gradient launch (FETCH_SIZE x 2 + WRITE_SIZE, separate passes) with and without the XCD-group placement
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
EAGER="--mode eager --steps 6 --warmup 3 --no-cpu-baseline --no-profile --no-parity --no-secondary"
rm -f $O/r06_pmc_dw_xcd.txt
for cfg in "0 0" "1 down" "1 up"; do
  set -- $cfg
  export MAGIC_DW_XCD_GROUPS=$1 MAGIC_SPLITK_POW2=$2
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pdw_c -- python3 $R/bench.py $EAGER > /dev/null 2> $O/r06_pdw_c.err || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pdw_d -- python3 $R/bench.py $EAGER > /dev/null 2> $O/r06_pdw_d.err || exit 1
  echo "MAGIC_DW_XCD_GROUPS=$1 MAGIC_SPLITK_POW2=$2" >> $O/r06_pmc_dw_xcd.txt
  python3 $R/profiles/pmc_traffic.py $O/pdw_c $O/pdw_d 9 | grep dw_batch >> $O/r06_pmc_dw_xcd.txt
  rm -rf $O/pdw_c $O/pdw_d
done
cat $O/r06_pmc_dw_xcd.txt
