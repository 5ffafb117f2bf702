cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/jtl
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/jtl -- python3 $R/tools/jump_probe.py 10000 16 > $R/gpurun_out/jtl.log 2>&1
tail -2 $R/gpurun_out/jtl.log
f=$(find $R/gpurun_out/jtl -name "*kernel_trace.csv" | head -1)
if [ -n "$f" ]; then python3 $R/tools/launch_list.py "$f" 300; fi
rm -rf $R/gpurun_out/jtl
