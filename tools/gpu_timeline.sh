# kernels and copies of the end of a run, in stream order:  gpurun -- 'bash tools/gpu_timeline.sh 24 tools/k1_once.py sdvpcmdecoder_amd/libsdvpcm_hip.so 3'
# columns: start (us, from the first event shown), gap to the event before, duration, name
N=$1; shift
R=$GRAFT_REPO_ROOT
tool=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tl
( cd $R && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/tl -- python3 $tool "$@" > gpurun_out/tl.log 2>&1 )
N=$N python3 - <<'PY'
import csv, glob, os
R = os.environ['GRAFT_REPO_ROOT']; N = int(os.environ['N'])
rows = []
for f in glob.glob(R + '/gpurun_out/tl/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0]))
for f in glob.glob(R + '/gpurun_out/tl/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)): rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'copy ' + r.get('Direction', '')))
rows.sort(); rows = rows[-N:]
t0 = rows[0][0]; prev = t0
for s, e, n in rows:
    print(f"{(s - t0) / 1e3:9.1f} {(s - prev) / 1e3:8.1f} {(e - s) / 1e3:9.1f}  {n}")
    prev = e
PY
rm -rf $R/gpurun_out/tl
