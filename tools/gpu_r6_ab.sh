cd $GRAFT_REPO_ROOT
for lib in "$@"; do
echo "== $lib"
SDVPCM_LIB=$lib timeout 300 python tools/pal_trace.py 2000 both 2>&1 | grep -v amdgpu | tail -1 | cut -c1-120
SDVPCM_LIB=$lib timeout 300 python tools/pal_trace.py 8000 lost 2>&1 | grep -v amdgpu | tail -1 | cut -c1-120
SDVPCM_LIB=$lib timeout 300 python tools/jump_probe.py 10000 16 2>&1 | grep -v amdgpu | tail -1
SDVPCM_LIB=$lib timeout 300 python tools/jump_trace.py bench 2>&1 | grep -v amdgpu | tail -1
done
