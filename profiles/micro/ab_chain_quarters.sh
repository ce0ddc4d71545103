#!/bin/bash
# round 6: the teacher's 32-row chain kernel with its FFN one quarter of the intermediate at a time (81 KB of LDS: a chain workgroup can share a CU with a 68 KB
# row-block backward workgroup of the student) against the whole GELU image (129 KB; vln-magic_amd/libmagic_hip_chainfull.so = the same tree built -DCHAIN_QUARTERS=0):
# student alone / teacher alone / together (profiles/micro/teacher_contention.py), then the step (bench.py), interleaved on one box
cd $GRAFT_REPO_ROOT
O=gpurun_out; rm -f $O/r06_ab_chain_quarters.txt
for rep in 1 2; do
for t in this chainfull; do
  if [ $t = this ]; then unset MAGIC_LIB_FILE MAGIC_ALLOW_STALE_LIB; else export MAGIC_LIB_FILE=$GRAFT_REPO_ROOT/vln-magic_amd/libmagic_hip_$t.so MAGIC_ALLOW_STALE_LIB=1; fi
  echo "== $t" | tee -a $O/r06_ab_chain_quarters.txt
  python profiles/micro/teacher_contention.py 2>/dev/null | grep "ms/step" | tee -a $O/r06_ab_chain_quarters.txt
  python bench.py --no-parity --no-secondary --no-cpu-baseline --no-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['ms_per_step'], d['steady']['ms_per_step'], d['steady']['ms_per_step_by_task'])" | tee -a $O/r06_ab_chain_quarters.txt
done
done
