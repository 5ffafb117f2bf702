cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_pcm1_vis.py tests/test_pcm1.py -m gpu -x -q 2>&1 | tail -2
timeout 300 python tools/pcm1_prof.py 10000 5 2>&1 | tail -3
