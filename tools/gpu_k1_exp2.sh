# round 5, experiment 2: one write burst per frame.  gpurun -- 'bash tools/gpu_k1_exp2.sh'
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_decode_frames.py tests/test_dropped_frames.py tests/test_sharded.py tests/test_stitch_kernel.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python tools/k1_ab.py 6 build/ab/r05_start.so build/variants/*.so 2>&1 | tail -12
