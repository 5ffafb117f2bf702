# PCM-16x0 back half with the libraries in build/variants/p16_*.so against the default build:  gpurun -- 'bash tools/gpu_p16_variants.sh'
cd $GRAFT_REPO_ROOT
for r in 1 2; do for v in "" $(ls build/variants/p16_*.so 2>/dev/null); do echo -n "${v:-default}: "; SDVPCM_LIB=$v timeout 300 python3 tools/pcm16_prof.py 10000 4 2>&1 | grep "it=3" | awk '{printf "%s %s ms  ", $1, $5}'; echo; done; done
