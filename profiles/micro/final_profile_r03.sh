#!/bin/bash
# round-3 measurement set (run through gpurun from the repo root): the default bench line, its rocprofv3 kernel stats (the exact
# driver command), the PMC traffic passes, the navigator-loop lines
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $R/gpurun_out/r03_bench.json 2> $R/gpurun_out/r03_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r03_prof -- python3 $R/bench.py --no-cpu-baseline --no-parity --no-secondary --no-profile > $R/gpurun_out/r03_bench_under_rocprof.json 2> $R/gpurun_out/r03_prof.err
find $R/gpurun_out/r03_prof -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/r03_kernel_stats.csv \;
find $R/gpurun_out/r03_prof -name "*kernel_trace.csv" -delete
rm -f $R/gpurun_out/r03_prof/*/*.db $R/gpurun_out/r03_prof/*.db
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --mode eager --steps 6 --warmup 3 --no-cpu-baseline --no-profile --no-parity --no-secondary > /dev/null 2> $R/gpurun_out/r03_pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --mode eager --steps 6 --warmup 3 --no-cpu-baseline --no-profile --no-parity --no-secondary > /dev/null 2> $R/gpurun_out/r03_pmc_write.err
python3 $R/profiles/pmc_traffic.py $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write 9 > $R/gpurun_out/r03_pmc_traffic.json
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
tail -c 400 $R/gpurun_out/r03_bench.json; echo; head -14 $R/gpurun_out/r03_pmc_traffic.json
python3 $R/bench_nav.py --steps 6 --warmup 2 > $R/gpurun_out/r03_bench_nav.json 2> $R/gpurun_out/r03_bench_nav.err
python3 $R/bench_nav.py --icod --hidden 128 --teacher-hidden 768 --instr-min 20 --instr-max 80 --hops-min 4 --hops-max 7 --max-action-len 15 --no-cpu-baseline --no-host-loop > $R/gpurun_out/r03_bench_nav_icod.json 2> $R/gpurun_out/r03_bench_nav_icod.err
tail -c 300 $R/gpurun_out/r03_bench_nav.json; echo; tail -c 300 $R/gpurun_out/r03_bench_nav_icod.json; echo
