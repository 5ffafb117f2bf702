# rocprofv3 kernel stats of the PCM-16x0 back half:  gpurun -- 'bash tools/gpu_pcm16_prof.sh [frames]'
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
N=${1:-4000}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_p16s
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_p16s -- python3 $R/tools/pcm16_prof.py $N 3 > $R/gpurun_out/prof_p16s.log 2>&1; echo "rocprof p16 stitch rc=$?"
grep "frames/s" $R/gpurun_out/prof_p16s.log
f=$(ls -t $R/gpurun_out/prof_p16s/*/*kernel_stats.csv 2>/dev/null | head -1); if [ -n "$f" ]; then head -12 "$f" | cut -d, -f1-8; fi
