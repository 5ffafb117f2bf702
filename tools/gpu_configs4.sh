# BASELINE configs[4] (one 100 000-frame tape, strong scaling) on the one GPU of the test box: 1 process, 1 RCCL rank, 2 and 8 gloo ranks sharing the GPU.
# gpurun -- 'bash tools/gpu_configs4.sh'      (the lines are kept as profiles/r05_configs4_*.json)
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
show() { python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); s = d.get("summary", {})
    print(sys.argv[2], {k: s.get(k) for k in ("configs4_frames", "configs4_ranks", "configs4_scaling", "configs4_ms", "configs4_frames_per_s", "configs4_frac", "configs4_process_group_ranks")})
except Exception as ex:
    print(sys.argv[2], "no line:", repr(ex))
PY
}
timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu 2> gpurun_out/c4_1proc.err | tail -1 > gpurun_out/configs4_1proc.json; echo "rc=$?"; show gpurun_out/configs4_1proc.json "one process:"
grep -o '"configs4_strong": {[^}]*}[^}]*}' gpurun_out/c4_1proc.err | tail -1 | cut -c1-1500
SDV_BENCH_FORCE_DIST=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29613 bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu 2> gpurun_out/c4_nccl1.err | tail -1 > gpurun_out/configs4_nccl_1rank.json; echo "rc=$?"; show gpurun_out/configs4_nccl_1rank.json "1 RCCL rank:"
SDV_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --frames 4000 --no-cpu 2> gpurun_out/c4_gloo2.err | tail -1 > gpurun_out/configs4_gloo_2ranks_one_gpu.json; echo "rc=$?"; show gpurun_out/configs4_gloo_2ranks_one_gpu.json "2 gloo ranks, one GPU:"
SDV_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29621 bench.py --gpus 8 --steps 3 --warmup 1 --frames 2000 --no-cpu 2> gpurun_out/c4_gloo8.err | tail -1 > gpurun_out/configs4_gloo_8ranks_one_gpu.json; echo "rc=$?"; show gpurun_out/configs4_gloo_8ranks_one_gpu.json "8 gloo ranks, one GPU:"
grep -h "Error\|error\|Traceback" gpurun_out/c4_*.err | head -5
