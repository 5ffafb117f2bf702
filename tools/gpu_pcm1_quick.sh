# quick GPU check of the PCM-1 back half: parity tests, timing, per-kernel profile
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_pcm1.py -m gpu -x -q 2>&1 | tail -3
python3 tools/pcm1_prof.py 10000 5 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_pcm1
( cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pcm1 -- python3 tools/pcm1_prof.py 10000 20 > gpurun_out/prof_pcm1.log 2>&1 )
cd $R
python3 - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/prof_pcm1/*/*kernel_stats.csv'))[-1]
for r in csv.DictReader(open(f)):
    if r['Name'].startswith('sdv_'):
        print(r['Name'][:44], r['Calls'], 'avg us', round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1), 'max', round(float(r['MaxNs'])/1e3,1))
PY
