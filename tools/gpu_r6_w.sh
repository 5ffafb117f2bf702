cd $GRAFT_REPO_ROOT
for lib in sdvpcmdecoder_amd/libsdvpcm_hip.so build/variants/w3.so build/variants/r5.so; do
  echo "== $lib"
  SDVPCM_LIB=$lib timeout 300 python tools/pal_trace.py 2000 both 2>&1 | grep -v amdgpu | tail -1
  SDVPCM_LIB=$lib timeout 300 python tools/pal_trace.py 8000 lost 2>&1 | grep -v amdgpu | tail -1
  SDVPCM_LIB=$lib timeout 300 python tools/pal_trace.py 8000 cells 2>&1 | grep -v amdgpu | tail -1
done
