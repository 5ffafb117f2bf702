# per-kernel times of the PCM-1 variants in build/variants/p1_*.so
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
for v in "" $(ls build/variants/p1_*.so 2>/dev/null); do
  echo "== variant ${v:-default}"
  SDVPCM_LIB=$v python3 tools/pcm1_prof.py 10000 5 2>&1 | tail -2
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/prof_pcm1
  ( cd $R && SDVPCM_LIB=$v rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pcm1 -- python3 tools/pcm1_prof.py 10000 20 > gpurun_out/prof_pcm1.log 2>&1 )
  cd $R
  python3 - <<'PY'
import csv,glob
f=sorted(glob.glob('gpurun_out/prof_pcm1/*/*kernel_stats.csv'))[-1]
for r in csv.DictReader(open(f)):
    if r['Name'].startswith('sdv_'):
        print(r['Name'][:44], r['Calls'], 'avg us', round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1), 'max', round(float(r['MaxNs'])/1e3,1))
PY
done
