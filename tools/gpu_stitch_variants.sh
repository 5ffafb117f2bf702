# stitch stage timing of the libraries in build/variants/st_*.so against the default build
cd $GRAFT_REPO_ROOT
for v in "" $(ls build/variants/st_*.so 2>/dev/null); do echo "== ${v:-default}"; SDVPCM_LIB=$v python3 tools/stitch_prof.py 10000 4 cont 2>&1 | tail -2; done
