# quick GPU check: binarize parity tests + bench line (no CPU baseline)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
python bench.py --steps 10 --warmup 2 --no-cpu --no-stitch 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('value %.0f frames/s  launch %.3f ms  frac %.3f' % (d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac']))"
