# PMC passes of the kernels VERDICT r02 asked counters for (one prefix per kernel family; gpurun -- 'bash tools/gpu_pmc_round3.sh')
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
bash $R/tools/gpu_pmc_kernel.sh 'sdv_k_stitch_(step|analyze)' stpmc $R/tools/stitch_prof.py 10000 2 cont
bash $R/tools/gpu_pmc_kernel.sh 'sdv_k_pcm16_analyse' p16spmc $R/tools/pcm16_prof.py 10000 1 si
bash $R/tools/gpu_pmc_kernel.sh 'sdv_k_pcm16_analyse' p16epmc $R/tools/pcm16_prof.py 10000 1 ei
bash $R/tools/gpu_pmc_kernel.sh 'sdv_k_pcm1_prescan|sdv_k_pcm1_frames_lean|sdv_k_pcm1_frames_bin' p1fpre $R/tools/pcm1_frames_prof.py 10000 1
bash $R/tools/gpu_pmc_kernel.sh 'sdv_k_pcm16_prescan|sdv_k_pcm16_frames_bin|sdv_k_pcm16_frames_lean' p16fpre $R/tools/pcm16_frames_prof.py 10000 1
bash $R/tools/gpu_pmc_kernel.sh 'sdv_k_ap_plan' applan $R/tools/audio_prof.py 10000 1
