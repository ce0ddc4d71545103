#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-profile > /dev/null 2> $R/gpurun_out/trace.err
f=$(find $R/gpurun_out/trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
print(len(rows), rows[0].keys())
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:50], r.get("Queue_Id", "")) for r in rows))
# last 8 steps = last ~8*400 kernels; take the final 30% of the trace
n = len(ev)
tail = ev[int(n * 0.55):]
t0, t1 = tail[0][0], max(e[1] for e in tail)
# union of busy intervals
busy, cur_s, cur_e = 0, None, None
for s, e, _, _ in tail:
    if cur_s is None: cur_s, cur_e = s, e
    elif s <= cur_e: cur_e = max(cur_e, e)
    else:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
busy += cur_e - cur_s
print("window ms", (t1 - t0) / 1e6, "busy ms", busy / 1e6, "idle frac", 1 - busy / (t1 - t0), "kernels", len(tail), "sum dur ms", sum(e - s for s, e, _, _ in tail) / 1e6)
# gap histogram between consecutive kernel starts vs previous end (any queue)
import collections
gaps = []
pe = tail[0][1]
for s, e, k, q in tail[1:]:
    if s > pe: gaps.append((s - pe, k))
    pe = max(pe, e)
gaps.sort(reverse=True)
print("n gaps", len(gaps), "total gap ms", sum(g for g, _ in gaps) / 1e6, "median gap us", gaps[len(gaps)//2][0] / 1e3)
print("largest gaps (us, next kernel):", [(round(g / 1e3, 1), k[:30]) for g, k in gaps[:12]])
qs = collections.Counter(q for _, _, _, q in tail)
print("queues", qs)
PY
rm -rf $R/gpurun_out/trace
