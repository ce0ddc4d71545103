#!/bin/bash
# per-call durations of one kernel family grouped by launch grid (which call sites are the slow ones): bash kernel_by_grid.sh ln_bwd
R=$GRAFT_REPO_ROOT; PAT=${1-ln_bwd}
cd /tmp && export TMPDIR=/tmp
[ -n "$LOCKSTEP" ] && export MAGIC_LOCKSTEP_EAGER=1      # LOCKSTEP=1: the paired launch structure of the captured graphs
rm -rf /tmp/kt && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python3 $R/bench.py --mode eager --steps 6 --warmup 3 --no-cpu-baseline --no-profile > /dev/null 2> /tmp/kt.err
python3 - "$PAT" <<'EOF'
import csv, glob, sys, collections
pat = sys.argv[1]
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/kt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            key = (r["Kernel_Name"][:48], "x".join(str(r.get(k, "?")) for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z")), r.get("Workgroup_Size_X", "?"))
            agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
    print(k, "n=%d avg=%.1f us min=%.1f max=%.1f" % (len(v), sum(v) / len(v), min(v), max(v)))
EOF
