mkdir -p gpurun_out/r05
NAV="--steps 10 --warmup 10 --no-cpu-baseline --no-host-loop --no-profile"
rm -f gpurun_out/r05/ab_kg_longk.txt
for r in 1 2 3; do
  for f in 0 1; do
    MAGIC_GEMM_KG_GROUP=$f timeout -k 10 200 python bench_nav.py $NAV 2>/dev/null | grep -o "\"ms_per_step\": [0-9.]*" | sed "s/^/nav  kg_group_longk=$f /" >> gpurun_out/r05/ab_kg_longk.txt || exit 1
  done
done
cat gpurun_out/r05/ab_kg_longk.txt
timeout -k 10 300 python -m pytest tests/test_nav_h768_oracle_gpu.py tests/test_step_graphs_gpu.py -m gpu -x -q > gpurun_out/r05/t_kgl.txt 2>&1; tail -1 gpurun_out/r05/t_kgl.txt
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r05/kglprof -- python3 $GRAFT_REPO_ROOT/bench_nav.py --steps 4 --warmup 8 --no-cpu-baseline --no-host-loop --no-profile > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r05/kglprof.err
cd $GRAFT_REPO_ROOT && find gpurun_out/r05/kglprof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r05/kgl_kernel_stats.csv \; ; rm -rf gpurun_out/r05/kglprof
grep "gemm_grouped" gpurun_out/r05/kgl_kernel_stats.csv | cut -c1-140
