cd $GRAFT_REPO_ROOT
SDV_STITCH_TIMING=1 python3 tools/stitch_prof.py 10000 2 2>&1 | tail -2
SDV_STITCH_TIMING=1 python3 tools/stitch_prof.py 256 2 2>&1 | tail -2
python -m pytest tests/test_stitch_kernel.py tests/test_deint_kernel.py -m gpu -x -q 2>&1 | tail -2
bash tools/gpu9.sh 2>&1 | tail -1
