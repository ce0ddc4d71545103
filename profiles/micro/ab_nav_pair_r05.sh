mkdir -p gpurun_out/r05
NAV="--steps 10 --warmup 10 --no-cpu-baseline --no-host-loop"
ICOD="--icod --hidden 128 --teacher-hidden 768 --instr-min 20 --instr-max 80 --hops-min 4 --hops-max 7 --max-action-len 15 --no-profile"
MAGIC_NAV_PAIR=1 timeout -k 10 200 python bench_nav.py $NAV > gpurun_out/r05/ab_nav_pair.json 2> gpurun_out/r05/ab_nav_pair.err &&
timeout -k 10 200 python bench_nav.py $NAV > gpurun_out/r05/ab_nav_plain.json 2> gpurun_out/r05/ab_nav_plain.err &&
timeout -k 10 200 python bench_nav.py $NAV $ICOD > gpurun_out/r05/ab_icod_plain.json 2> gpurun_out/r05/ab_icod_plain.err &&
MAGIC_LANE_PROBE=1 timeout -k 10 200 python bench_nav.py $NAV $ICOD > gpurun_out/r05/ab_icod_probe.json 2> gpurun_out/r05/ab_icod_probe.err &&
MAGIC_NAV_PAIR=1 timeout -k 10 200 python bench_nav.py $NAV $ICOD > gpurun_out/r05/ab_icod_pair.json 2> gpurun_out/r05/ab_icod_pair.err &&
MAGIC_NAV_PAIR=1 timeout -k 10 300 python -m pytest tests/test_nav_h768_oracle_gpu.py tests/test_step_graphs_gpu.py -m gpu -x -q > gpurun_out/r05/t_pair.txt 2>&1
tail -3 gpurun_out/r05/t_pair.txt
