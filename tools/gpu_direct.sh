# the fused entry under two builds, same box:  gpurun -- 'bash tools/gpu_direct.sh a b'
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_decode_frames.py tests/test_sharded.py -m gpu -x -q 2>&1 | tail -2
for v in $1 $2 $1 $2; do
  echo "== $v"; SDVPCM_LIB=build/variants/$v.so python tools/fused_prof.py 10000 8 | tail -3
done
SDVPCM_LIB=$GRAFT_REPO_ROOT/build/variants/$2.so bash tools/gpu_fused_timeline.sh 2>&1 | grep -v "rocprofv3"
