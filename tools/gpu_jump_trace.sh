# scheduler trace of the window-jumps tape of bench.py (developer build):  gpurun -- 'bash tools/gpu_jump_trace.sh'
cd $GRAFT_REPO_ROOT
SDVPCM_LIB=build/variants/dev.so timeout 300 python tools/jump_trace.py bench 2>&1 | grep "sched\] iter\|jumps\|^[0-9]" | cut -c1-220 | tail -40
