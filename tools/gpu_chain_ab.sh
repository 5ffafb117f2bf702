cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do for v in "" $(ls build/variants/*.so 2>/dev/null); do echo -n "variant ${v:-product}: "; SDVPCM_LIB=$v python bench.py --steps 5 --warmup 2 --no-cpu 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1]); s=d['stitch_stage']
print('launch %.3f stitch %.3f frames_to_pcm %.3f masked_pcm %.3f' % (d['roofline']['avg_launch_ms'], s['stitch_ms_per_step'], s['frames_to_pcm_ms_per_step'], s['frames_to_masked_pcm_ms_per_step']))"; done; done
