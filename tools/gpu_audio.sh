# AudioProcessor stage on the GPU box:  gpurun -- 'bash tools/gpu_audio.sh'
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_audio.py -m gpu -x -q 2>&1 | tail -25
python tools/audio_prof.py 10000 4 cpu 2>&1 | grep -v "^\[AP\]\|startTimer\|\[TA\]" | tee gpurun_out/audio_prof.log
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_audio
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_audio -- python3 $R/tools/audio_prof.py 10000 5 > $R/gpurun_out/prof_audio.log 2>&1; echo "rocprof audio rc=$?"
python3 - <<'PY'
import csv, glob, os
fs = sorted(glob.glob(os.environ['GRAFT_REPO_ROOT'] + '/gpurun_out/prof_audio/*/*kernel_stats.csv'), key=os.path.getmtime)
for r in csv.DictReader(open(fs[-1])):
    if r['Name'].startswith('sdv_'):
        print(r['Name'], 'calls', r['Calls'], 'avg ns', r['AverageNs'], 'min', r['MinNs'], 'max', r['MaxNs'])
PY
