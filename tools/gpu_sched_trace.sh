cd $GRAFT_REPO_ROOT
SDV_SCHED_TRACE=1 SDVPCM_LIB=build/variants/dev.so timeout 300 python tools/pal_trace.py 2000 both 2>&1 | grep -v amdgpu | tail -120
