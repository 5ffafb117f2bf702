# PMC passes and kernel statistics of the kernels round 4 added / reworked for damaged tapes (gpurun -- 'bash tools/gpu_pmc_round4.sh')
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
timeout 300 bash $R/tools/gpu_pmc_kernel.sh 'sdv_k_stc007_frames\(' fullpmc $R/tools/pal_trace.py 2000 both
# (round 6: the builds the C3 tape's rounds run in - the general build without snapshots for the big rounds, the five-wave one for the small)
timeout 300 bash $R/tools/gpu_pmc_kernel.sh 'sdv_k_stc007_frames_plain' plainpmc $R/tools/pal_trace.py 2000 both
timeout 300 bash $R/tools/gpu_pmc_kernel.sh 'sdv_k_stc007_frames_fat' fatpmc $R/tools/pal_trace.py 2000 both
timeout 300 bash $R/tools/gpu_pmc_kernel.sh 'sdv_k_stc007_sweep_levels' swlpmc $R/tools/pal_trace.py 2000 both
timeout 300 bash $R/tools/gpu_pmc_kernel.sh 'sdv_k_stc007_sweep_pick' swppmc $R/tools/pal_trace.py 2000 both
timeout 300 bash $R/tools/gpu_pmc_kernel.sh 'sdv_k_hist_carry' hcpmc $R/tools/jump_probe.py 10000 16
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_c3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -- python3 $R/tools/pal_trace.py 2000 both > $R/gpurun_out/prof_c3.log 2>&1; echo "rocprof c3 rc=$?"
