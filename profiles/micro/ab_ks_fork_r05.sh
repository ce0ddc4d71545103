mkdir -p gpurun_out/r05
NAV="--steps 10 --warmup 10 --no-cpu-baseline --no-host-loop --no-profile"
rm -f gpurun_out/r05/ab_ks_fork.txt
for r in 1 2 3; do
  for f in 0 1; do
    MAGIC_LOCKSTEP_FORK=$f timeout -k 10 200 python bench_nav.py $NAV 2>gpurun_out/r05/ab_ks_fork.err | grep -o "\"ms_per_step\": [0-9.]*" | sed "s/^/nav  ks_fork=$f /" >> gpurun_out/r05/ab_ks_fork.txt || exit 1
  done
done
cat gpurun_out/r05/ab_ks_fork.txt
timeout -k 10 300 python -m pytest tests/test_nav_h768_oracle_gpu.py tests/test_step_graphs_gpu.py tests/test_glue_gpu.py -m gpu -x -q > gpurun_out/r05/t_ks_fork.txt 2>&1; tail -2 gpurun_out/r05/t_ks_fork.txt
