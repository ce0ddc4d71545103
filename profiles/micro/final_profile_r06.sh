#!/bin/bash
# round-6 measurement set (run through gpurun from the repo root): the default bench line, its rocprofv3 kernel stats (the exact driver
# command), four PMC passes (MFMA busy / LDS + wave cycles / FETCH_SIZE / WRITE_SIZE; never combined with a trace option), navigator lines.
#   bash profiles/micro/final_profile_r06.sh [bench|prof|pmc|nav ...]     (default: all)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
WHAT=${@:-bench prof pmc nav navprof}
EAGER="--mode eager --steps 6 --warmup 3 --no-cpu-baseline --no-profile --no-parity --no-secondary"
for w in $WHAT; do
case $w in
bench)
  python3 $R/bench.py > $O/r06_bench.json 2> $O/r06_bench.err || exit 1
  tail -c 600 $O/r06_bench.json; echo ;;
prof)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06_prof -- python3 $R/bench.py --no-cpu-baseline --no-parity --no-secondary --no-profile > $O/r06_bench_under_rocprof.json 2> $O/r06_prof.err || exit 1
  find $O/r06_prof -name "*kernel_stats.csv" -exec cp {} $O/r06_kernel_stats.csv \;
  find $O/r06_prof -name "*kernel_trace.csv" -delete
  rm -f $O/r06_prof/*/*.db $O/r06_prof/*.db ;;
pmc)
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_a -- python3 $R/bench.py $EAGER > /dev/null 2> $O/r06_pmc_a.err || exit 1
  echo pass A done
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/pmc_b -- python3 $R/bench.py $EAGER > /dev/null 2> $O/r06_pmc_b.err || exit 1
  echo pass B done
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_c -- python3 $R/bench.py $EAGER > /dev/null 2> $O/r06_pmc_c.err || exit 1
  echo pass C done
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_d -- python3 $R/bench.py $EAGER > /dev/null 2> $O/r06_pmc_d.err || exit 1
  echo pass D done
  STATS=$O/r06_kernel_stats.csv; [ -f $STATS ] || STATS=$R/profiles/r06_rocprofv3_kernel_stats.csv      # (a `prof` leg of this call, else the committed one)
  python3 $R/profiles/pmc_kernels.py $STATS $O/pmc_a $O/pmc_b $O/pmc_c $O/pmc_d > $O/r06_pmc_kernels.json
  python3 $R/profiles/pmc_traffic.py $O/pmc_c $O/pmc_d 9 > $O/r06_pmc_traffic.json
  rm -rf $O/pmc_a $O/pmc_b $O/pmc_c $O/pmc_d
  head -c 1500 $O/r06_pmc_kernels.json; echo ;;
nav)
  python3 $R/bench_nav.py --steps 10 --warmup 10 > $O/r06_bench_nav.json 2> $O/r06_bench_nav.err
  python3 $R/bench_nav.py --icod --hidden 128 --teacher-hidden 768 --instr-min 20 --instr-max 80 --hops-min 4 --hops-max 7 --max-action-len 15 --steps 10 --warmup 10 --no-cpu-baseline --no-host-loop > $O/r06_bench_nav_icod.json 2> $O/r06_bench_nav_icod.err
  tail -c 300 $O/r06_bench_nav.json; echo; tail -c 300 $O/r06_bench_nav_icod.json; echo ;;
navpmc)
  # the navigator iteration's kernels under the same four counter passes (graph replays included: the paired / grouped launches are what runs)
  NAVP="--steps 2 --warmup 6 --no-cpu-baseline --no-host-loop --no-profile"
  timeout -k 10 280 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/npmc_a -- python3 $R/bench_nav.py $NAVP > /dev/null 2> $O/r06_npmc_a.err || exit 1
  echo nav pass A done
  timeout -k 10 280 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/npmc_b -- python3 $R/bench_nav.py $NAVP > /dev/null 2> $O/r06_npmc_b.err || exit 1
  echo nav pass B done
  timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/npmc_c -- python3 $R/bench_nav.py $NAVP > /dev/null 2> $O/r06_npmc_c.err || exit 1
  timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/npmc_d -- python3 $R/bench_nav.py $NAVP > /dev/null 2> $O/r06_npmc_d.err || exit 1
  echo nav passes C D done
  STATS=$O/r06_kernel_stats_nav.csv; [ -f $STATS ] || STATS=$R/profiles/r06_rocprofv3_kernel_stats_nav.csv
  python3 $R/profiles/pmc_kernels.py $STATS $O/npmc_a $O/npmc_b $O/npmc_c $O/npmc_d > $O/r06_pmc_kernels_nav.json
  rm -rf $O/npmc_a $O/npmc_b $O/npmc_c $O/npmc_d
  head -c 1200 $O/r06_pmc_kernels_nav.json; echo ;;
navprof)
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06_navprof -- python3 $R/bench_nav.py --steps 4 --warmup 8 --no-cpu-baseline --no-host-loop --no-profile > $O/r06_bench_nav_under_rocprof.json 2> $O/r06_navprof.err || exit 1
  find $O/r06_navprof -name "*kernel_stats.csv" -exec cp {} $O/r06_kernel_stats_nav.csv \;
  rm -rf $O/r06_navprof
  # kernels per steady iteration as the profiler counts them (graph replays included; the instrumented pass of bench_nav.py launches eagerly and
  # cannot see the paired step graphs): the same command with 8 more timed iterations, the difference of the two dispatch counts / 8
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/r06_navprof2 -- python3 $R/bench_nav.py --steps 12 --warmup 8 --no-cpu-baseline --no-host-loop --no-profile > /dev/null 2> $O/r06_navprof2.err || exit 1
  find $O/r06_navprof2 -name "*kernel_stats.csv" -exec cp {} $O/r06_kernel_stats_nav12.csv \;
  rm -rf $O/r06_navprof2
  python3 - $O/r06_kernel_stats_nav.csv $O/r06_kernel_stats_nav12.csv > $O/r06_nav_launches.txt <<'PY'
import csv, sys
def tot(p):
    rows = list(csv.DictReader(open(p)))
    return sum(int(r["Calls"]) for r in rows), sum(float(r["TotalDurationNs"]) for r in rows)
(c4, t4), (c12, t12) = tot(sys.argv[1]), tot(sys.argv[2])
print(f"rocprofv3 --kernel-trace --stats of bench_nav.py (MAGIC-L navigator iteration), 8 warm-up + 4 timed iterations: {c4} kernel dispatches, {t4 / 1e6:.1f} ms of kernel time;")
print(f"8 warm-up + 12 timed: {c12} dispatches, {t12 / 1e6:.1f} ms.  Per steady iteration (difference / 8): {(c12 - c4) / 8:.0f} kernel dispatches, {(t12 - t4) / 8e6:.1f} ms of kernel time")
PY
  cat $O/r06_nav_launches.txt
  python3 $R/profiles/micro/nav_kernel_breakdown.py --graphs --iters 8 > $O/r06_nav_breakdown.txt 2>&1
  MAGIC_NAV_TIMERS=1 python3 $R/profiles/micro/nav_kernel_breakdown.py --graphs --iters 6 2>&1 | grep -A12 "^host sections" > $O/r06_nav_host_sections.txt
  python3 $R/profiles/micro/nav_kernel_breakdown.py --graphs --icod --iters 8 2>&1 | grep "^iteration\|^instrumented" > $O/r06_nav_breakdown_icod.txt
  grep "^iteration" $O/r06_nav_breakdown.txt | tail -4 ;;
esac
done
