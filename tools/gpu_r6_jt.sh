cd $GRAFT_REPO_ROOT
SDVPCM_LIB=build/variants/dev.so timeout 300 python tools/jump_trace.py bench > gpurun_out/jump_trace_bench.log 2>&1
grep -v 'link \|amdgpu' gpurun_out/jump_trace_bench.log | grep 'iter\|jumps\|moved on\|carried' | tail -40
