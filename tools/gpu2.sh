set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --frames 2000 --steps 3 --warmup 1 --cpu-frames 300 > gpurun_out/bench_small.json 2> gpurun_out/bench_small.err; echo "rc=$?"; cat gpurun_out/bench_small.json; tail -5 gpurun_out/bench_small.err
python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; echo "rc=$?"; cat gpurun_out/bench_full.json; tail -5 gpurun_out/bench_full.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_bench.err; echo "rc=$?"
cat $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json
find $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -name '*stats*' | head
