# PMC counters of the PCM-1 frame kernel (instruction mix, HBM traffic); separate passes as the microarchitecture guide prescribes
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/p1pmc1 $R/gpurun_out/p1pmc2 $R/gpurun_out/p1pmc3 $R/gpurun_out/p1pmc4
rocprofv3 --kernel-include-regex 'sdv_k_pcm1_frames' --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/p1pmc1 -- python3 $R/tools/pcm1_prof.py 10000 2 > /dev/null 2> $R/gpurun_out/p1pmc1.err; echo "pmc1 rc=$?"
rocprofv3 --kernel-include-regex 'sdv_k_pcm1_frames' --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/p1pmc2 -- python3 $R/tools/pcm1_prof.py 10000 2 > /dev/null 2> $R/gpurun_out/p1pmc2.err; echo "pmc2 rc=$?"
rocprofv3 --kernel-include-regex 'sdv_k_pcm1_frames' --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p1pmc3 -- python3 $R/tools/pcm1_prof.py 10000 2 > /dev/null 2> $R/gpurun_out/p1pmc3.err; echo "pmc3 rc=$?"
rocprofv3 --kernel-include-regex 'sdv_k_pcm1_frames' --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p1pmc4 -- python3 $R/tools/pcm1_prof.py 10000 2 > /dev/null 2> $R/gpurun_out/p1pmc4.err; echo "pmc4 rc=$?"
cd $R
for d in p1pmc1 p1pmc2 p1pmc3 p1pmc4; do python3 tools/pmc_summary.py gpurun_out/$d; done
