# kernel statistics of one profiling tool under one build:  gpurun -- 'bash tools/gpu_kstats.sh build/variants/x.so tools/pcm1_frames_prof.py 10000 3'
R=$GRAFT_REPO_ROOT
export SDVPCM_LIB=$R/$1; shift
tool=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/kstats
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/kstats -- python3 $tool "$@" > $R/gpurun_out/kstats.log 2>&1; echo "rc=$?"
f=$(ls -t $R/gpurun_out/kstats/*/*kernel_stats.csv 2>/dev/null | head -1); head -12 "$f" | cut -d, -f1-8
rm -rf $R/gpurun_out/kstats
