# the C3 PAL tape through several builds of the library on one box, alternating:  gpurun -- 'bash tools/gpu_ab_c3.sh lib1.so lib2.so ...'
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for lib in "$@"; do
    echo -n "$(basename $lib): "; SDVPCM_LIB=$lib timeout 300 python tools/pal_trace.py 2000 both 2>&1 | tail -1
  done
done
