#!/bin/bash
# final measurement set of the round (run through gpurun from the repo root)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $R/gpurun_out/bench_final.json 2> $R/gpurun_out/bench_final.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_final -- python3 $R/bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-profile > $R/gpurun_out/bench_under_rocprof.json 2> $R/gpurun_out/prof_final.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/bench.py --mode eager --steps 6 --warmup 3 --no-cpu-baseline --no-profile > /dev/null 2> $R/gpurun_out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/bench.py --mode eager --steps 6 --warmup 3 --no-cpu-baseline --no-profile > /dev/null 2> $R/gpurun_out/pmc_write.err
python3 $R/profiles/pmc_traffic.py $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write 9 > $R/gpurun_out/pmc_traffic.json
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_mfma -- python3 $R/bench.py --mode eager --steps 6 --warmup 3 --no-cpu-baseline --no-profile > /dev/null 2> $R/gpurun_out/pmc_mfma.err
python3 $R/profiles/pmc_mfma.py $R/gpurun_out/pmc_mfma > $R/gpurun_out/pmc_mfma.json; rm -rf $R/gpurun_out/pmc_mfma
find $R/gpurun_out/prof_final -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/kernel_stats_final.csv \;
# keep the merge-back small: drop the raw traces
rm -rf $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
find $R/gpurun_out/prof_final -name "*kernel_trace.csv" -delete
tail -c 600 $R/gpurun_out/bench_final.json; echo; cat $R/gpurun_out/pmc_traffic.json | head -12
# navigator loop (SURVEY f-1): bench line + kernel stats
cd /tmp
python3 $R/bench_nav.py --steps 6 --warmup 2 > $R/gpurun_out/bench_nav_final.json 2> $R/gpurun_out/bench_nav_final.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_nav -- python3 $R/bench_nav.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-loop --no-profile > $R/gpurun_out/bench_nav_under_rocprof.json 2> $R/gpurun_out/prof_nav.err
find $R/gpurun_out/prof_nav -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/kernel_stats_nav.csv \;
find $R/gpurun_out/prof_nav -name "*kernel_trace.csv" -delete
rm -f $R/gpurun_out/prof_nav/*.db
tail -c 300 $R/gpurun_out/bench_nav_final.json; echo
