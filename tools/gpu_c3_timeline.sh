cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/c3tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/c3tl -- python3 $R/tools/pal_trace.py 2000 both > $R/gpurun_out/c3tl.log 2>&1
tail -2 $R/gpurun_out/c3tl.log
f=$(find $R/gpurun_out/c3tl -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then python3 $R/tools/c3_timeline.py "$f"; fi
rm -rf $R/gpurun_out/c3tl
