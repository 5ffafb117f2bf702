# the sharded bench loop with 8 ranks (gloo) sharing the one GPU of the test box: exercises the 8-way split of the tape
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
SDV_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29621 bench.py --gpus 8 --steps 3 --warmup 1 --frames 2000 --no-cpu 2> gpurun_out/bench_8rank.err | tail -1 > gpurun_out/bench_8rank_gloo.json; echo "rc=$?"
cut -c1-900 gpurun_out/bench_8rank_gloo.json; grep -v "^\[W\|Warning\|return func\|amdgpu.ids\|^$" gpurun_out/bench_8rank.err | tail -5
