# where a turn of sdv_k_stitch_step spends its cycles (developer build):  gpurun -- 'bash tools/gpu_step_timing.sh'
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_stitch_kernel.py tests/test_decode_frames.py -m gpu -x -q 2>&1 | tail -2
SDVPCM_LIB=build/variants/dev.so SDV_STITCH_TIMING=1 timeout 300 python tools/stitch_prof.py 10000 3 cont 2>&1 | grep -i "timing" | tail -2
timeout 300 python tools/stitch_prof.py 10000 5 cont 2>&1 | tail -3
timeout 300 python tools/fused_prof.py 2>&1 | tail -4
