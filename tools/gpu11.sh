cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/stprof
( cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stprof -- python3 tools/stitch_prof.py 10000 3 ) 2>&1 | tail -5
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/stprof -name "*kernel_stats.csv" | head -1); cat $f | cut -d, -f1-8 | head -12
