mkdir -p gpurun_out/r05
NAV="--steps 10 --warmup 10 --no-cpu-baseline --no-host-loop --no-profile"
rm -f gpurun_out/r05/ab_plan_ahead.txt
for r in 1 2 3; do
  for pa in off inline; do
    timeout -k 10 200 python bench_nav.py $NAV --plan-ahead $pa 2>gpurun_out/r05/ab_plan_ahead.err | grep -o "\"ms_per_step\": [0-9.]*" | sed "s/^/nav  plan_ahead=$pa /" >> gpurun_out/r05/ab_plan_ahead.txt || exit 1
  done
done
cat gpurun_out/r05/ab_plan_ahead.txt
