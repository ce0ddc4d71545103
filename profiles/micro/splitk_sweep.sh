mkdir -p gpurun_out
for cfg in "128 8" "128 16" "64 8" "256 8" "256 16" "192 12" "512 8"; do
  set -- $cfg
  r=$(MAGIC_SPLITK_TARGET=$1 MAGIC_SPLITK_MIN_TILES=$2 timeout -k 10 120 python bench.py --no-cpu-baseline --no-parity --no-secondary --steps 60 --warmup 12 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['detail']['kernels_launches_per_step_and_avg_us']['magic_gemm_dw_grouped'])")
  echo "target=$1 min_tiles=$2 ms_per_step=$r" | tee -a gpurun_out/splitk_sweep.log
done
