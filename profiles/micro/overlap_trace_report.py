"""report for overlap_trace.py: python3 profiles/micro/overlap_trace_report.py <kernel_trace.csv>"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
# split at pauses >= 40 ms; the timed phases are the 2nd, 4th, 6th bursts from the END (warm-up burst, pause, timed burst, pause)
bursts, cur = [], [ev[0]]
for e in ev[1:]:
    if e[0] - max(x[1] for x in cur[-50:]) > 40e6:
        bursts.append(cur)
        cur = []
    cur.append(e)
bursts.append(cur)
print("bursts:", [len(b) for b in bursts])
timed = bursts[-5], bursts[-3], bursts[-1]


def short(k):
    k = k.replace("void ", "")
    for tok in ("_kernel", "Kernel"):
        i = k.find(tok)
        if i > 0:
            k = k[:i]
    return k[-40:]


stats = {}
for name, b in zip("abc", timed):
    span = (max(e[1] for e in b) - b[0][0]) / 1e6
    print(f"\n=== phase {name}: {len(b)} kernels, span {span:.2f} ms = {span / 36:.3f} ms/step")
    byq = collections.defaultdict(list)
    for e in b:
        byq[e[3]].append(e)
    for q, es in sorted(byq.items(), key=lambda kv: -len(kv[1])):
        dur = sum(e[1] - e[0] for e in es) / 1e6
        gaps = [max(0, es[i][0] - es[i - 1][1]) for i in range(1, len(es))]
        gaps_small = [g for g in gaps if g < 100e3]
        print(f"  queue {q}: {len(es)} kernels ({len(es) / 36:.1f}/step), kernel time {dur / 36 * 1e3:.0f} us/step, gaps<100us {sum(gaps_small) / 36 / 1e3:.0f} us/step "
              f"(mean {sum(gaps_small) / max(len(gaps_small), 1) / 1e3:.2f} us), larger gaps {sum(g for g in gaps if g >= 100e3) / 36 / 1e3:.0f} us/step")
        per = collections.defaultdict(lambda: [0, 0])
        for e in es:
            p = per[short(e[2])]
            p[0] += 1
            p[1] += e[1] - e[0]
        stats[(name, q)] = per
# per-kernel mean durations, phase a vs alone, for the names that matter
qa = sorted({q for (n, q) in stats if n == "a"}, key=lambda q: -sum(v[0] for v in stats[("a", q)].values()))
for q in qa:
    alone = None
    for (n, q2), per in stats.items():
        if n in "bc" and q2 == q:
            alone = per
    if alone is None:
        continue
    print(f"\nqueue {q}: mean kernel duration overlapped vs alone (us), calls/step")
    tot_o = tot_a = 0
    for k, (c, t) in sorted(stats[("a", q)].items(), key=lambda kv: -kv[1][1])[:24]:
        if k in alone and alone[k][0]:
            o, a = t / c / 1e3, alone[k][1] / alone[k][0] / 1e3
            tot_o += t / 36 / 1e3
            tot_a += alone[k][1] / 36 / 1e3
            print(f"  {k:42s} {o:7.1f} {a:7.1f}  x{o / a:4.2f}  {c / 36:5.1f}")
    print(f"  listed kernels: {tot_o:.0f} us/step overlapped vs {tot_a:.0f} alone")
