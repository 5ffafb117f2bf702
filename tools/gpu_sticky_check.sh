cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_decode_frames.py -m gpu -x -q 2>&1 | tail -2
timeout 300 python tools/jump_probe.py 10000 16 2>&1 | tail -2
timeout 300 python tools/pal_trace.py 2000 both 2>&1 | tail -1
timeout 600 python tools/k1_ab.py 5 build/ab/wavesync.so sdvpcmdecoder_amd/libsdvpcm_hip.so 2>&1 | tail -2
