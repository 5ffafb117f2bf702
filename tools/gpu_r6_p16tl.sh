export SDVPCM_LIB=$GRAFT_REPO_ROOT/build/variants/p16a.so
bash $GRAFT_REPO_ROOT/tools/gpu_p16_timeline.sh
