#!/bin/bash
# step time of the headline bench with the frozen teacher's stream confined to N compute units (MAGIC_TEACHER_CUS, host/trainer._side_stream)
out=gpurun_out/cu_mask_sweep.txt
: > $out
for spec in "" 192:spread 128:spread 96:spread 64:spread 128:low 128:xcd 64:xcd; do
  MAGIC_TEACHER_CUS=$spec timeout -k 10 300 python bench.py --no-cpu-baseline --no-profile --no-parity --no-secondary --steps 60 --warmup 12 2> gpurun_out/cu_mask_err.txt \
    | python -c "import sys, json; d = json.loads(sys.stdin.readlines()[-1]); print('teacher CUs %-12s %.4f ms/step' % ('$spec' or 'all', d['ms_per_step']))" >> $out || { echo "failed at $spec" >> $out; tail -5 gpurun_out/cu_mask_err.txt >> $out; break; }
done
cat $out
