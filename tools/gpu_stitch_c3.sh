cd $GRAFT_REPO_ROOT
timeout 300 python tools/stitch_c3_prof.py 2000 3
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_stc3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_stc3 -- python3 $GRAFT_REPO_ROOT/tools/stitch_c3_prof.py 2000 3 > /dev/null 2>&1
f=$(find $GRAFT_REPO_ROOT/gpurun_out/prof_stc3 -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then grep "stitch\|Name" "$f" | cut -c1-160; fi
find $GRAFT_REPO_ROOT/gpurun_out/prof_stc3 -name "*kernel_trace.csv" -delete
