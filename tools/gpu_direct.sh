# the fused entry with and without "direct fields", same box:  gpurun -- 'bash tools/gpu_direct.sh'
cd $GRAFT_REPO_ROOT
for v in head direct2 head direct2; do
  echo "== $v"; SDVPCM_LIB=$GRAFT_REPO_ROOT/build/variants/$v.so bash tools/gpu_fused_timeline.sh 2>&1 | grep -v "rocprofv3\|copyBuffer"
  SDVPCM_LIB=build/variants/$v.so python tools/fused_prof.py 10000 8 | tail -2
done
