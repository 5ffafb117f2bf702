cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for f in si ei; do
rm -rf $R/gpurun_out/prof_p16_$f
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_p16_$f -- python3 $R/tools/pcm16_prof.py 10000 3 $f > $R/gpurun_out/prof_p16_$f.log 2>&1
echo "== $f"; grep "it=2" $R/gpurun_out/prof_p16_$f.log
python3 - <<PY
import csv, glob, os
fs = sorted(glob.glob("$R/gpurun_out/prof_p16_$f/*/*kernel_stats.csv"), key=os.path.getmtime)
for r in csv.DictReader(open(fs[-1])):
    if r['Name'].startswith('sdv_k_pcm16'):
        print("  %-28s calls %4s avg %9.1f us  total per run %7.2f ms" % (r['Name'].split('(')[0], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/3e6))
PY
done
