cd $GRAFT_REPO_ROOT
SDV_STITCH_TIMING=1 python3 tools/stitch_prof.py 10000 2 2>&1 | tail -4
SDV_STITCH_TIMING=1 python3 tools/stitch_prof.py 256 2 2>&1 | tail -4
