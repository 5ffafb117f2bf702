# one profiling tool under several builds, same box:  gpurun -- 'bash tools/gpu_ab_tool.sh "tools/pcm1_frames_prof.py 10000 3" a.so b.so ...'
cd $GRAFT_REPO_ROOT
cmd=$1; shift
for v in "$@"; do echo "== $v"; SDVPCM_LIB=$v timeout 300 python $cmd 2>&1 | grep "mode 2\|it=\|frames/s" | tail -2; done
