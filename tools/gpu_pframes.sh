# PCM-1 / PCM-16x0 frame drivers: parity + A/B of two builds.  gpurun -- 'bash tools/gpu_pframes.sh old.so new.so'
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
SDVPCM_LIB=$2 timeout 900 python -m pytest tests/test_pcm16_frames.py tests/test_pcm16_front.py tests/test_pcm1_frames.py tests/test_pcm1_front.py tests/test_decode_frames.py -m gpu -x -q 2>&1 | tail -3
for v in "$@" "$@"; do
  echo "== $v"
  SDVPCM_LIB=$v timeout 300 python tools/pcm1_frames_prof.py 10000 3 2>&1 | grep "mode 2"
  SDVPCM_LIB=$v timeout 300 python tools/pcm16_frames_prof.py 10000 3 2>&1 | grep "mode 1\|mode 2"
done
