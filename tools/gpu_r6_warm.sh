cd $GRAFT_REPO_ROOT
for w in 5 5 50 50 5 200; do
python bench.py --gpus 1 --steps 20 --warmup $w --no-stitch --no-cpu 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('warmup', d['warmup'], 'ms_per_step %.4f' % d['ms_per_step'], 'launch %.4f' % d['roofline']['avg_launch_ms'])"
done
