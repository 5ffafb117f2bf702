# quick GPU check of what was touched last:  gpurun -- 'bash tools/gpu_quick.sh'
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_decode_frames.py tests/test_sharded.py -m gpu -x -q 2>&1 | tail -3
timeout 900 python tools/soak.py 12 7000 2>&1 | tail -2
for i in 1 2 3; do python bench.py --steps 5 --warmup 2 --no-cpu 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().split('\n')[-1]); s = d['summary']
print('ms_per_step', round(d['ms_per_step'], 4), 'launch', round(d['roofline']['avg_launch_ms'], 4), 'e2e', round(s['end_to_end_ms_per_step'], 3), 'jumps', round(s['damaged_window_jumps_ms_per_step'], 3), 'lost', round(s['damaged_lost_lines_ms_per_step'], 3), 'c3 bin', round(s['pal_damaged_binarize_ms_per_step'], 2))"; done
