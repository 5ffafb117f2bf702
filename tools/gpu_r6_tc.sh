cd $GRAFT_REPO_ROOT
for what in both lost cells; do
  SDVPCM_LIB=build/variants/dev.so timeout 300 python tools/pal_trace.py 2000 $what 2>&1 | grep -v amdgpu | tail -2
  SDV_NO_TC=1 SDVPCM_LIB=build/variants/dev.so timeout 300 python tools/pal_trace.py 2000 $what 2>&1 | grep -v amdgpu | tail -1
done
