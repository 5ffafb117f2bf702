# round 6: the window-jump tape and the C3 tape after a scheduler change (product library: timings; developer build: trace)
cd $GRAFT_REPO_ROOT
timeout 300 python tools/jump_probe.py 10000 16 2>&1 | grep -v amdgpu | tail -3
timeout 300 python tools/pal_trace.py 2000 both 2>&1 | grep -v amdgpu | tail -3
timeout 300 python tools/rung_probe.py 10000 2>&1 | grep -v amdgpu | tail -6
python tools/k1_ab.py 8 build/variants/r5.so sdvpcmdecoder_amd/libsdvpcm_hip.so 2>&1 | tail -4
SDVPCM_LIB=build/variants/dev.so SDV_SCHED_TRACE=1 timeout 300 python tools/jump_probe.py 10000 16 > gpurun_out/jump_trace.log 2>&1
SDVPCM_LIB=build/variants/dev.so SDV_SCHED_TRACE=1 timeout 300 python tools/pal_trace.py 2000 both > gpurun_out/pal_trace_both.log 2>&1
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
