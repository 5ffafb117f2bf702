# exercises the RCCL (backend "nccl") plumbing of the sharded bench loop with the one GPU of the test box: one rank, real collectives
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
SDV_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu --no-stitch 2> gpurun_out/bench_nccl_1rank.err | tail -1 > gpurun_out/bench_nccl_1rank.json; echo "rc=$?"
cut -c1-700 gpurun_out/bench_nccl_1rank.json; tail -5 gpurun_out/bench_nccl_1rank.err
SDV_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --frames 4000 --no-cpu 2> gpurun_out/bench_2rank.err | tail -1 | cut -c1-300; echo "2-rank gloo rc=$?"
python bench.py --steps 3 --warmup 1 --no-cpu | cut -c1-200
