"""The frame-kernel launches of the last call in a rocprofv3 --kernel-trace csv: kernel, workgroups, duration.  usage: launch_list.py <kernel_trace.csv> [min gap between calls, us]"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    gx = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))); wx = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)))
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], gx // max(1, wx)))
rows.sort()
gap = (int(sys.argv[2]) if len(sys.argv) > 2 else 400) * 1000
calls, cur = [], [rows[0]]
for a, b in zip(rows[:-1], rows[1:]):
    if b[0] - a[1] > gap:
        calls.append(cur); cur = []
    cur.append(b)
calls.append(cur)
calls = [c for c in calls if sum(1 for k in c if k[2].startswith("sdv_k_stc007_frames")) >= 3]
c = calls[-1]
t0 = c[0][0]
print(f"last call: {len(c)} kernels over {(c[-1][1] - t0) / 1e6:.2f} ms")
for s, e, n, g in c:
    if n.startswith("sdv_k_stc007") or n == "sdv_k_hist_carry":
        print(f"  +{(s - t0) / 1e3:9.1f} us  {n:32s} {g:6d} workgroups  {(e - s) / 1e3:8.1f} us")
