# developer aid: per-phase cycle stamps of the PCM-1 frame kernel, then the usual parity + timing
cd $GRAFT_REPO_ROOT
SDV_STITCH_TIMING=1 python3 tools/pcm1_prof.py 10000 2 2>&1 | tail -2
bash tools/gpu_pcm1_quick.sh
