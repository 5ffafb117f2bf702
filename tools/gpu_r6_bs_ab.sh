cd $GRAFT_REPO_ROOT
for lib in "$@"; do
echo "== $lib"
SDVPCM_LIB=$lib timeout 300 python tools/pal_trace.py 2000 both 2>&1 | grep -v amdgpu | tail -1 | cut -c1-140
SDVPCM_LIB=$lib timeout 300 python tools/pal_trace.py 8000 cells 2>&1 | grep -v amdgpu | tail -1 | cut -c1-140
SDVPCM_LIB=$lib timeout 300 python tools/pal_trace.py 200 cells 2>&1 | grep -v amdgpu | tail -1 | cut -c1-140
done
