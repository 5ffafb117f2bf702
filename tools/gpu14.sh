cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 5 --warmup 1 2>&1 | tail -3 | tee gpurun_out/bench_r01b.json
