cd $GRAFT_REPO_ROOT
SDVPCM_LIB=build/variants/stamps.so timeout 300 python tools/slow_stamps.py 2000 both 2>&1 | grep -v "first 40\|amdgpu"
SDVPCM_LIB=build/variants/stamps.so timeout 300 python tools/slow_stamps.py 10000 jumps 2>&1 | grep -v "first 40\|amdgpu"
