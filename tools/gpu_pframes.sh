# round 5: PCM-1 / PCM-16x0 frame drivers, lean builds.  gpurun -- 'bash tools/gpu_pframes.sh'
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_pcm1_frames.py tests/test_pcm16_frames.py tests/test_pcm1_front.py tests/test_pcm16_front.py tests/test_decode_frames.py tests/test_dropped_frames.py tests/test_pcm1.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3
for v in build/ab/r05_start.so build/variants/plean.so build/variants/plean_w3.so; do
  echo "== $v"
  SDVPCM_LIB=$v timeout 300 python tools/pcm1_frames_prof.py 10000 3 2>&1 | grep "mode 2"
  SDVPCM_LIB=$v timeout 300 python tools/pcm16_frames_prof.py 10000 3 2>&1 | grep "mode 2"
done
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_p1f $R/gpurun_out/prof_p16f
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_p1f -- python3 $R/tools/pcm1_frames_prof.py 10000 3 > $R/gpurun_out/prof_p1f.log 2>&1; echo "rocprof p1f rc=$?"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_p16f -- python3 $R/tools/pcm16_frames_prof.py 10000 3 > $R/gpurun_out/prof_p16f.log 2>&1; echo "rocprof p16f rc=$?"
for d in prof_p1f prof_p16f; do f=$(ls -t $R/gpurun_out/$d/*/*kernel_stats.csv 2>/dev/null | head -1); if [ -n "$f" ]; then head -8 "$f" | cut -d, -f1-8; fi; done
