cd $GRAFT_REPO_ROOT
python bench.py 2>gpurun_out/bench_stderr.log | tail -1 > gpurun_out/bench_line.json; cat gpurun_out/bench_line.json | cut -c1-1500
