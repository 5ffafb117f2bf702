# stitch stage timing of the libraries in build/variants/st_*.so against the default build:  gpurun -- 'bash tools/gpu_stitch_variants.sh'
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for v in "" $(ls build/variants/st_*.so 2>/dev/null); do echo -n "${v:-default}: "; SDVPCM_LIB=$v timeout 300 python3 tools/stitch_prof.py 10000 6 cont 2>&1 | tail -3 | awk '{printf "%s/%s ", $4, $7}'; echo; done; done
