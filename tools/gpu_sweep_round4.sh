# round 4: the sweep kernels - parity on the GPU, the C3 tape, kernel statistics
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
python tools/pal_trace.py 2000 both 2>&1 | tail -5
python tools/pal_trace.py 2000 cells 2>&1 | tail -2
python tools/jump_probe.py 2>&1 | tail -6
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r04/c3prof -- python3 $GRAFT_REPO_ROOT/tools/pal_trace.py 2000 both > $GRAFT_REPO_ROOT/gpurun_out/r04/c3prof.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r04/c3prof -name "*kernel_stats.csv" | head -1); echo $f; head -12 $f | cut -c1-200
