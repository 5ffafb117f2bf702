# quick GPU check of the stitch stage: parity tests + timing of continuing 10 000-frame batches with per-phase cycle stamps
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_stitch_kernel.py tests/test_deint_kernel.py -m gpu -x -q 2>&1 | tail -2
SDV_STITCH_TIMING=1 python3 tools/stitch_prof.py 10000 4 cont 2>&1 | tail -3
