# damaged tapes and the PCM frame drivers through several builds of the library on one box:  gpurun -- 'bash tools/gpu_ab_tapes.sh lib1.so lib2.so ...'
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in "$@"; do
    echo "== $(basename $lib)"
    SDVPCM_LIB=$lib timeout 300 python tools/jump_probe.py 10000 16 2>&1 | tail -1
    SDVPCM_LIB=$lib timeout 300 python tools/pal_trace.py 2000 both 2>&1 | tail -1
    SDVPCM_LIB=$lib timeout 300 python tools/pcm1_frames_prof.py 10000 3 2>&1 | tail -1
    SDVPCM_LIB=$lib timeout 300 python tools/pcm16_frames_prof.py 10000 3 2>&1 | tail -1
  done
done
