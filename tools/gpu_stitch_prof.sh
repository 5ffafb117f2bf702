# rocprofv3 kernel stats of the stitch stage (continuing 10 000-frame batches)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_stitch
( cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stitch -- python3 tools/stitch_prof.py 10000 5 cont > gpurun_out/prof_stitch.log 2>&1 )
cd $R
python3 - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/prof_stitch/*/*kernel_stats.csv'))[-1]
for r in csv.DictReader(open(f)):
    if r['Name'].startswith('sdv_'):
        print(r['Name'][:44], r['Calls'], 'avg us', round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1), 'max', round(float(r['MaxNs'])/1e3,1))
PY
