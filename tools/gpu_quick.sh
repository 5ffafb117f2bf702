# quick check of a change in the damaged-tape path: parity of the frame kernels, the two damaged tapes
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
timeout 300 python tools/pal_trace.py 2000 both 2>&1 | tail -4
timeout 300 python tools/jump_probe.py 10000 16 2>&1 | tail -2
