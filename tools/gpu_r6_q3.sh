cd $GRAFT_REPO_ROOT
for lib in build/variants/rb4.so; do
echo "== $lib"
SDVPCM_LIB=$lib timeout 300 python tools/pal_trace.py 2000 both 2>&1 | grep -v amdgpu | tail -1
SDVPCM_LIB=$lib timeout 300 python tools/pal_trace.py 8000 cells 2>&1 | grep -v amdgpu | tail -1
done
