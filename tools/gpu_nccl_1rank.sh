# the sharded path on the one GPU of the test box: GPU tests of the C++ example, one RCCL rank, two gloo ranks:  gpurun -- 'bash tools/gpu_nccl_1rank.sh'
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_sharded.py -m gpu -x -q 2>&1 | tail -2
SDV_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu 2> gpurun_out/bench_nccl_1rank.err | tail -1 > gpurun_out/bench_nccl_1rank.json; echo "rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_nccl_1rank.json').read()); print('1 RCCL rank: sharded_full_path_ms', d.get('summary',{}).get('sharded_full_path_ms'), 'one engine, same cold file', d.get('summary',{}).get('sharded_one_engine_same_file_ms'), 'steady fused', d.get('summary',{}).get('end_to_end_ms_per_step'))
PY
SDV_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --frames 4000 --no-cpu 2> gpurun_out/bench_2rank.err | tail -1 > gpurun_out/bench_2rank_gloo_one_gpu.json; echo "2-rank gloo rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_2rank_gloo_one_gpu.json').read()); print('2 gloo ranks on one GPU: sharded_full_path_ms', d.get('summary',{}).get('sharded_full_path_ms'))
PY
