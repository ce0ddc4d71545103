mkdir -p gpurun_out/r05
NAV="--steps 10 --warmup 10 --no-cpu-baseline --no-host-loop --no-profile"
rm -f gpurun_out/r05/ab_fcat_nav.txt
for r in 1 2 3; do
  for f in 0 1; do
    MAGIC_DW_FLUSH_CAT=$f timeout -k 10 200 python bench_nav.py $NAV 2>gpurun_out/r05/ab_fcat_nav.err | grep -o "\"ms_per_step\": [0-9.]*" | sed "s/^/nav  flush_cat=$f /" >> gpurun_out/r05/ab_fcat_nav.txt || exit 1
  done
done
cat gpurun_out/r05/ab_fcat_nav.txt
