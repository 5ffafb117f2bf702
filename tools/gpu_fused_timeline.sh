# kernel timeline of one steady fused call:  gpurun -- 'bash tools/gpu_fused_timeline.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/ftl
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/ftl -- python3 $R/tools/fused_prof.py > $R/gpurun_out/ftl.log 2>&1
tail -2 $R/gpurun_out/ftl.log
python3 - <<'PY'
import csv, glob, os
R = os.environ['GRAFT_REPO_ROOT']
rows = []
for f in glob.glob(R + '/gpurun_out/ftl/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0]))
for f in glob.glob(R + '/gpurun_out/ftl/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'copy ' + r.get('Direction', '') + ' ' + r.get('Bytes', r.get('Size', ''))))
rows.sort()
# the last call: from the last sdv_k_predict on
idx = max(i for i, r in enumerate(rows) if r[2] == 'sdv_k_predict')
t0 = rows[idx][0]; prev = t0
for s, e, n in rows[idx:]:
    print(f"{(s - t0) / 1e3:9.1f} {(s - prev) / 1e3:8.1f} {(e - s) / 1e3:9.1f}  {n}")
    prev = e
PY
rm -rf $R/gpurun_out/ftl
