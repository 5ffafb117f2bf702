# quick GPU check of the PCM-1 back half: parity tests, timing, per-kernel profile
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_pcm1.py -m gpu -x -q 2>&1 | tail -3
python3 tools/pcm1_prof.py 10000 5 2>&1 | tail -5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/pcm1_prof -o pcm1 -- python3 $GRAFT_REPO_ROOT/tools/pcm1_prof.py 10000 5 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(ls -t gpurun_out/pcm1_prof/*/*kernel_stats.csv gpurun_out/pcm1_prof/*kernel_stats.csv 2>/dev/null | head -1)
echo "stats: $f"; head -8 "$f"
