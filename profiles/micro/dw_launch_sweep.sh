#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; rm -f $O/r06_dw_launch_exp.txt
P="python profiles/micro/dw_launch_probe.py"
export MAGIC_DW_XCD_GROUPS=0 MAGIC_SPLITK_POW2=0
{
$P --label "product" || exit 1
$P --label "product" --shared-operands
for t in "$@"; do
MAGIC_LIB_FILE=$GRAFT_REPO_ROOT/vln-magic_amd/libmagic_hip_$t.so MAGIC_ALLOW_STALE_LIB=1 $P --label "$t"
MAGIC_LIB_FILE=$GRAFT_REPO_ROOT/vln-magic_amd/libmagic_hip_$t.so MAGIC_ALLOW_STALE_LIB=1 $P --label "$t" --shared-operands
done
} 2>&1 | grep -v amdgpu.ids | tee $O/r06_dw_launch_exp.txt
