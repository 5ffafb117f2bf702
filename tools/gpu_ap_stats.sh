# what the audio plan kernel spends its time on (developer build with -DSDV_AP_STATS in build/variants/apstats.so):  gpurun -- 'bash tools/gpu_ap_stats.sh'
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_audio.py -m gpu -x -q 2>&1 | tail -2
SDVPCM_LIB=build/variants/apstats.so timeout 300 python tools/audio_prof.py 10000 2 2>&1 | grep "ap plan" | tail -3
timeout 300 python tools/audio_prof.py 10000 5 2>&1 | grep "it=4"
