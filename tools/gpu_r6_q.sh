cd $GRAFT_REPO_ROOT
lib=${1:-build/variants/rb.so}
echo "== $lib"
SDVPCM_LIB=$lib timeout 300 python tools/rung_probe.py 10000 0,-2,3,4,-4 2>&1 | grep -v amdgpu | tail -5
SDVPCM_LIB=$lib timeout 300 python tools/jump_probe.py 10000 16 2>&1 | grep -v amdgpu | tail -1
SDVPCM_LIB=$lib timeout 300 python tools/jump_trace.py bench 2>&1 | grep -v amdgpu | tail -1
SDVPCM_LIB=$lib timeout 300 python tools/pal_trace.py 2000 both 2>&1 | grep -v amdgpu | tail -1
SDVPCM_LIB=$lib timeout 300 python tools/pal_trace.py 8000 lost 2>&1 | grep -v amdgpu | tail -1
python tools/k1_ab.py 8 build/variants/r5.so $lib 2>&1 | tail -2
