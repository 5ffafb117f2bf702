cd $GRAFT_REPO_ROOT
SDVPCM_LIB=build/variants/stamps.so timeout 300 python tools/slow_stamps.py 2000 both 2>&1 | tail -14
SDVPCM_LIB=build/variants/stamps.so timeout 300 python tools/slow_stamps.py 2000 cells 2>&1 | tail -14
