# round 6, first look: the tree as it stands (smoke, GPU suite, bench line) and the scheduler traces of the two damaged tapes (developer build)
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
python bench.py 2>/dev/null | tail -1 | cut -c1-2500
SDVPCM_LIB=build/variants/dev.so SDV_SCHED_TRACE=1 timeout 300 python tools/pal_trace.py 2000 both > gpurun_out/pal_trace_both.log 2>&1
SDVPCM_LIB=build/variants/dev.so SDV_SCHED_TRACE=1 timeout 300 python tools/jump_probe.py 10000 16 > gpurun_out/jump_trace.log 2>&1
tail -4 gpurun_out/pal_trace_both.log; tail -3 gpurun_out/jump_trace.log
