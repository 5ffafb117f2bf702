cd $GRAFT_REPO_ROOT
for v in "" build/variants/st_w4.so build/variants/st_w5.so; do echo "== ${v:-default}"; SDVPCM_LIB=$v python3 tools/stitch_prof.py 10000 4 cont 2>&1 | tail -2; done
