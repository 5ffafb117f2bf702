# round 5: where the damaged tapes stand.  gpurun -- 'bash tools/gpu_damaged.sh'
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 300 python tools/pal_trace.py 2000 both 2>&1 | tail -5
timeout 300 python tools/jump_probe.py 10000 16 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/c3tl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/c3tl -- python3 $R/tools/pal_trace.py 2000 both > $R/gpurun_out/c3tl.log 2>&1
f=$(find $R/gpurun_out/c3tl -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then python3 $R/tools/c3_timeline.py "$f" | tail -60; fi
rm -rf $R/gpurun_out/c3tl
