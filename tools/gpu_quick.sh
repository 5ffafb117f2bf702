# quick GPU check of what was touched last:  gpurun -- 'bash tools/gpu_quick.sh'
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_pcm16.py tests/test_pcm16_vis.py tests/test_pcm16_asm.py tests/test_decode_frames.py tests/test_sharded.py -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/pcm16_prof.py 10000 4 2>&1 | grep "it=[0-3]"
