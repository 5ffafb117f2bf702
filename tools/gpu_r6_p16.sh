cd $GRAFT_REPO_ROOT
for lib in build/variants/p16c.so build/variants/p16d.so; do
echo "== $lib"
SDVPCM_LIB=$lib timeout 300 python tools/pcm16_prof.py 10000 5 si,ei 2>&1 | grep 'it=[34]' | cut -c1-60
done
