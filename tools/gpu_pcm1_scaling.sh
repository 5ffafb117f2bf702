# frames-kernel time against batch size: separates per-wave latency from throughput limits
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
for n in 500 1000 2500 5000 10000 20000 40000; do
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/prof_pcm1
  ( cd $R && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pcm1 -- python3 tools/pcm1_prof.py $n 10 > gpurun_out/prof_pcm1.log 2>&1 )
  cd $R
  python3 - $n <<'PY'
import csv,glob,sys
f=sorted(glob.glob('gpurun_out/prof_pcm1/*/*kernel_stats.csv'))[-1]
for r in csv.DictReader(open(f)):
    if r['Name'].startswith('sdv_k_pcm1_frames'):
        print('n', sys.argv[1], r['Name'][:20], 'avg us', round(float(r['AverageNs'])/1e3,1), 'min', round(float(r['MinNs'])/1e3,1))
PY
done
