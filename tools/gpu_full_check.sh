cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 1500 python tools/soak.py 24 7000 > gpurun_out/soak_ldsctx.log 2>&1; echo "soak rc=$?"; tail -2 gpurun_out/soak_ldsctx.log
timeout 300 python tools/pal_trace.py 2000 both 2>&1 | tail -2
timeout 300 python tools/pal_trace.py 2000 cells 2>&1 | tail -1
timeout 300 python tools/pal_trace.py 2000 lost 2>&1 | tail -1
