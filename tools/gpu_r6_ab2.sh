cd $GRAFT_REPO_ROOT
bash tools/gpu_r6_ab.sh build/variants/r5.so build/variants/cur.so
echo "== curdev, no snapshots"
SDV_NO_TC=1 SDVPCM_LIB=build/variants/curdev.so timeout 300 python tools/pal_trace.py 2000 both 2>&1 | grep -v amdgpu | tail -1 | cut -c1-120
SDV_NO_TC=1 SDVPCM_LIB=build/variants/curdev.so timeout 300 python tools/pal_trace.py 8000 lost 2>&1 | grep -v amdgpu | tail -1 | cut -c1-120
echo "== curdev"
SDVPCM_LIB=build/variants/curdev.so timeout 300 python tools/pal_trace.py 2000 both 2>&1 | grep -v amdgpu | tail -1 | cut -c1-120
