# A/B of the audio stage between the product library and build/variants/*.so:  gpurun -- 'bash tools/gpu_audio_ab.sh'
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do for v in "" $(ls build/variants/*.so 2>/dev/null); do echo -n "variant ${v:-product}: "; SDVPCM_LIB=$v python tools/audio_prof.py 10000 6 2>/dev/null | grep "clean.*it=5\|worn.*it=5" | awk '{printf "%s %s ms  ", $1, $5}'; echo; done; done
