cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_decode_frames.py tests/test_dropped_frames.py -m gpu -x -q 2>&1 | tail -2
for v in build/ab/r05_start.so build/variants/plean.so sdvpcmdecoder_amd/libsdvpcm_hip.so; do
  echo "== $v"
  SDVPCM_LIB=$v timeout 300 python tools/pal_trace.py 2000 both 2>&1 | tail -1
  SDVPCM_LIB=$v timeout 300 python tools/jump_probe.py 10000 16 2>&1 | tail -1
done
