"""Timeline of the last call of tools/pal_trace.py from a rocprofv3 --kernel-trace csv: GPU-busy time, gaps between kernels, the largest gaps.
usage: c3_timeline.py <kernel_trace.csv>"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]))
rows.sort()
# the calls are separated by the host work of the tool (synchronize + print): gaps over 1.5 ms
calls, cur = [], [rows[0]]
for a, b in zip(rows[:-1], rows[1:]):
    if b[0] - a[1] > 1_500_000:
        calls.append(cur); cur = []
    cur.append(b)
calls.append(cur)
calls = [c for c in calls if any(k[2].startswith("sdv_k_stc007_frames") for k in c)]
c = calls[-1]
span = c[-1][1] - c[0][0]
busy = sum(e - s for s, e, _ in c)
print(f"last call: {len(c)} kernels, span {span / 1e6:.2f} ms, GPU busy {busy / 1e6:.2f} ms, gaps {(span - busy) / 1e6:.2f} ms")
per = {}
for s, e, n in c:
    per.setdefault(n, [0, 0]); per[n][0] += 1; per[n][1] += e - s
for n, (k, t) in sorted(per.items(), key=lambda x: -x[1][1]):
    print(f"  {n:34s} {k:4d} launches {t / 1e6:8.3f} ms")
gaps = sorted(((b[0] - a[1], a[2], b[2]) for a, b in zip(c[:-1], c[1:])), reverse=True)
print("largest gaps (us, after, before):", [(g // 1000, x, y) for g, x, y in gaps[:12]])
hist = {}
for g, x, y in gaps:
    hist.setdefault((x, y), [0, 0]); hist[(x, y)][0] += 1; hist[(x, y)][1] += g
print("gaps by kernel pair (count, total us):", sorted(((v[0], v[1] // 1000, k) for k, v in hist.items()), key=lambda t: -t[1])[:10])
