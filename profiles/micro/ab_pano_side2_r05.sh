mkdir -p gpurun_out/r05
NAV="--steps 10 --warmup 10 --no-cpu-baseline --no-host-loop --no-profile"
ICOD="--icod --hidden 128 --teacher-hidden 768 --instr-min 20 --instr-max 80 --hops-min 4 --hops-max 7 --max-action-len 15"
rm -f gpurun_out/r05/ab_pano_side2.txt
for r in 1 2; do
  for ps in 0 teacher student; do
    MAGIC_PANO_BWD_STREAM=$ps timeout -k 10 200 python bench_nav.py $NAV $ICOD 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*" | sed "s/^/icod pano_side=$ps /" >> gpurun_out/r05/ab_pano_side2.txt || exit 1
  done
done
cat gpurun_out/r05/ab_pano_side2.txt
