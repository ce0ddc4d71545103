"""timeline of a rocprofv3 kernel trace (csv): per iteration-sized window, the union of kernel intervals (GPU busy), the idle time, and the sum of
kernel durations (which counts concurrent kernels twice).  Iterations are cut at the AdamW launches (multi_tensor_apply / adamw).
  python3 profiles/micro/trace_timeline.py <kernel_trace.csv> [marker substring]"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
rows.sort()
marker = sys.argv[2] if len(sys.argv) > 2 else "multi_tensor_apply"
cuts = []
last = None
for i, (s, e, n, q) in enumerate(rows):
    if marker in n:
        if last is None or s - last > 20e6:      # first optimizer launch of an iteration
            cuts.append(i)
        last = s
print(f"{len(rows)} dispatches, {len(cuts)} iterations (cut at '{marker}')")
for a, b in zip(cuts[:-1], cuts[1:]):
    seg = rows[a:b]
    t0, t1 = seg[0][0], seg[-1][1]
    busy = 0
    cs, ce = seg[0][0], seg[0][1]
    gaps = []
    for s, e, n, q in seg[1:]:
        if s > ce:
            busy += ce - cs
            gaps.append(s - ce)
            cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    tot = sum(e - s for s, e, _, _ in seg)
    big = sum(g for g in gaps if g > 20e3)
    # where does the backward start: the first attn_bwd / ln_bwd launch
    tb = next((s for s, e, n, q in seg if "bwd" in n), t1)
    fwd_busy = sum(min(e, tb) - s for s, e, n, q in seg if s < tb)
    print(f"window {(t1 - t0) / 1e6:7.1f} ms: busy {busy / 1e6:6.1f}  idle {(t1 - t0 - busy) / 1e6:6.1f} (gaps > 20 us: {big / 1e6:5.1f} ms in {sum(g > 20e3 for g in gaps)})  "
          f"sum of durations {tot / 1e6:6.1f}  launches {len(seg)} | until first backward kernel {(tb - t0) / 1e6:6.1f} ms (kernel time in it {fwd_busy / 1e6:5.1f})")
