cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_stitch_kernel.py tests/test_decode_frames.py tests/test_render.py -m gpu -x -q 2>&1 | tail -2
SDVPCM_LIB=build/variants/dev.so SDV_STITCH_TIMING=1 timeout 300 python tools/stitch_prof.py 10000 3 cont 2>&1 | grep -i "analyze timing" | tail -1
timeout 300 python tools/stitch_prof.py 10000 5 cont 2>&1 | tail -3
