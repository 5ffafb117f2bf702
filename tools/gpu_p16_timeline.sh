# kernel timeline of the last SI call of tools/pcm16_prof.py:  gpurun -- 'bash tools/gpu_p16_timeline.sh'
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/ptl
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/ptl -- python3 $R/tools/pcm16_prof.py 10000 3 si > $R/gpurun_out/ptl.log 2>&1
tail -1 $R/gpurun_out/ptl.log
python3 - <<'PY'
import csv, glob, os
R = os.environ['GRAFT_REPO_ROOT']
rows = []
for f in glob.glob(R + '/gpurun_out/ptl/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0]))
for f in glob.glob(R + '/gpurun_out/ptl/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'copy'))
rows.sort()
idx = max(i for i, r in enumerate(rows) if r[2] == 'sdv_k_pcm16_segments' and (i == 0 or rows[i][0] - rows[i - 1][1] > 200000))
t0 = rows[idx][0]
for s, e, n in rows[idx:idx + 70]:
    print(f"{(s - t0) / 1e3:9.1f} .. {(e - t0) / 1e3:9.1f}  {(e - s) / 1e3:8.1f}  {n}")
PY
rm -rf $R/gpurun_out/ptl
