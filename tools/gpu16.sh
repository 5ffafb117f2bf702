cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python bench.py --steps 5 --warmup 1 2>&1 | tail -1 | cut -c1-400
echo "--- 2 ranks on one GPU (gloo), sharded tape"
SDV_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --frames 3000 --no-cpu 2>&1 | tail -2 | cut -c1-900
