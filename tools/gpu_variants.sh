# times the libraries in build/variants/ (tuning experiments) with the bench's steady-state loop
cd $GRAFT_REPO_ROOT
for v in "" $(ls build/variants/*.so 2>/dev/null); do
  echo -n "variant ${v:-default}: "
  SDVPCM_LIB=$v python bench.py --steps 10 --warmup 2 --no-cpu --no-stitch 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('value %.0f frames/s  launch %.3f ms  frac %.3f  words_ok %s' % (d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['config']['decoded_words_match_generator']))"
done
