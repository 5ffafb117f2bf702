cd $GRAFT_REPO_ROOT
for v in "" build/variants/st_a4.so; do
  echo "== variant: ${v:-default}"
  SDV_STITCH_TIMING=1 SDVPCM_LIB=$v python3 tools/stitch_prof.py 256 2 2>&1 | tail -2 | head -1
done
