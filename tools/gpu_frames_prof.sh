# rocprofv3 kernel stats of the PCM-1 and PCM-16x0 frame drivers:  gpurun -- 'bash tools/gpu_frames_prof.sh'
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_p1f $R/gpurun_out/prof_p16f
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_p1f -- python3 $R/tools/pcm1_frames_prof.py 2000 3 > $R/gpurun_out/prof_p1f.log 2>&1; echo "rocprof p1f rc=$?"
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_p16f -- python3 $R/tools/pcm16_frames_prof.py 2000 3 > $R/gpurun_out/prof_p16f.log 2>&1; echo "rocprof p16f rc=$?"
for d in prof_p1f prof_p16f; do f=$(ls -t $R/gpurun_out/$d/*/*kernel_stats.csv 2>/dev/null | head -1); if [ -n "$f" ]; then head -8 "$f" | cut -d, -f1-8; fi; done
