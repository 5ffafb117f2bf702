# Full validation + measurement pass on the GPU box:  gpurun -- 'bash tools/gpu_validate.sh'
# smoke, GPU parity suite, bench (1 GPU and 2 gloo ranks on the one GPU), rocprofv3 kernel stats, PMC passes of the hot kernel.
# Results land in gpurun_out/ (copied to profiles/ by hand when they are the ones to keep).
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1800 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py > gpurun_out/bench_line.json 2> gpurun_out/bench_full.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/bench_line.json
cp gpurun_out/bench_details.json gpurun_out/bench_full.json      # the full objects of this run (later bench runs of this script write the file again)
SDV_BENCH_FORCE_DIST=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu 2> gpurun_out/bench_nccl1.err | tail -1 > gpurun_out/bench_nccl_1rank.json; echo "nccl 1-rank rc=$?"
SDV_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 1 --frames 4000 --no-cpu 2> gpurun_out/bench_2rank.err | tail -1 > gpurun_out/bench_2rank_gloo.json; echo "2-rank rc=$?"
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_bench $R/gpurun_out/prof_stitch $R/gpurun_out/pmc1 $R/gpurun_out/pmc2 $R/gpurun_out/pmc3 $R/gpurun_out/pmc4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_bench -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof_bench.err; echo "rocprof bench rc=$?"
# ... and of the headline leg alone (no other leg launches the lean kernel there: its average is the launch duration the roofline line quotes)
rm -rf $R/gpurun_out/prof_headline
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_headline -- python3 $R/bench.py --steps 20 --warmup 2 --no-cpu --no-stitch > $R/gpurun_out/prof_headline.json 2> $R/gpurun_out/prof_headline.err; echo "rocprof headline rc=$?"
( cd $R && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stitch -- python3 tools/stitch_prof.py 10000 5 cont > gpurun_out/prof_stitch.log 2>&1 ); echo "rocprof stitch rc=$?"
timeout 600 rocprofv3 --kernel-include-regex 'sdv_k_stc007_frames_lean' --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc1 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-stitch > /dev/null 2> $R/gpurun_out/pmc1.err; echo "pmc1 rc=$?"
timeout 600 rocprofv3 --kernel-include-regex 'sdv_k_stc007_frames_lean' --pmc SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/pmc2 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-stitch > /dev/null 2> $R/gpurun_out/pmc2.err; echo "pmc2 rc=$?"
timeout 600 rocprofv3 --kernel-include-regex 'sdv_k_stc007_frames_lean' --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc3 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-stitch > /dev/null 2> $R/gpurun_out/pmc3.err; echo "pmc3 rc=$?"
timeout 600 rocprofv3 --kernel-include-regex 'sdv_k_stc007_frames_lean' --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc4 -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu --no-stitch > /dev/null 2> $R/gpurun_out/pmc4.err; echo "pmc4 rc=$?"
# PCM-1 back half: kernel stats and PMC passes of its frame kernel
rm -rf $R/gpurun_out/prof_pcm1 $R/gpurun_out/p1pmc1 $R/gpurun_out/p1pmc2 $R/gpurun_out/p1pmc3 $R/gpurun_out/p1pmc4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_pcm1 -- python3 $R/tools/pcm1_prof.py 10000 20 > $R/gpurun_out/prof_pcm1.log 2>&1; echo "rocprof pcm1 rc=$?"
timeout 600 rocprofv3 --kernel-include-regex 'sdv_k_pcm1_frames' --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/p1pmc1 -- python3 $R/tools/pcm1_prof.py 10000 2 > /dev/null 2> $R/gpurun_out/p1pmc1.err; echo "p1pmc1 rc=$?"
timeout 600 rocprofv3 --kernel-include-regex 'sdv_k_pcm1_frames' --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/p1pmc2 -- python3 $R/tools/pcm1_prof.py 10000 2 > /dev/null 2> $R/gpurun_out/p1pmc2.err; echo "p1pmc2 rc=$?"
timeout 600 rocprofv3 --kernel-include-regex 'sdv_k_pcm1_frames' --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p1pmc3 -- python3 $R/tools/pcm1_prof.py 10000 2 > /dev/null 2> $R/gpurun_out/p1pmc3.err; echo "p1pmc3 rc=$?"
timeout 600 rocprofv3 --kernel-include-regex 'sdv_k_pcm1_frames' --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p1pmc4 -- python3 $R/tools/pcm1_prof.py 10000 2 > /dev/null 2> $R/gpurun_out/p1pmc4.err; echo "p1pmc4 rc=$?"
# PCM-1 front half: kernel stats and PMC passes of its line kernel
bash $R/tools/gpu_pcm1_front_pmc.sh 2>&1 | grep "rc="

# PCM-1 and PCM-16x0 frame drivers: kernel stats
bash $R/tools/gpu_frames_prof.sh 2>&1 | grep "rc="
# PCM-16x0 back half: kernel stats
bash $R/tools/gpu_pcm16_prof.sh 10000 2>&1 | grep "rc=\|frames/s"
# AudioProcessor stage: kernel stats over the three tapes of tools/audio_prof.py
rm -rf $R/gpurun_out/prof_audio
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_audio -- python3 $R/tools/audio_prof.py 10000 5 > $R/gpurun_out/prof_audio.log 2>&1; echo "rocprof audio rc=$?"
grep "it=4" $R/gpurun_out/prof_audio.log
# ... and the HBM traffic of its streaming pass (sdv_k_ap_prepare: input -> the caller's buffer + bitmaps), PMC passes on the clean tape
rm -rf $R/gpurun_out/apmc3 $R/gpurun_out/apmc4
timeout 600 rocprofv3 --kernel-include-regex 'sdv_k_ap_prepare' --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/apmc3 -- python3 $R/tools/audio_prof.py 10000 1 > /dev/null 2> $R/gpurun_out/apmc3.err; echo "apmc3 rc=$?"
timeout 600 rocprofv3 --kernel-include-regex 'sdv_k_ap_prepare' --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/apmc4 -- python3 $R/tools/audio_prof.py 10000 1 > /dev/null 2> $R/gpurun_out/apmc4.err; echo "apmc4 rc=$?"
# visualiser canvases: kernel stats
rm -rf $R/gpurun_out/prof_vis
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_vis -- python3 $R/tools/vis_prof.py 2000 4 > $R/gpurun_out/prof_vis.log 2>&1; echo "rocprof vis rc=$?"
tail -1 $R/gpurun_out/prof_vis.log
# PMC passes of the stitch kernels, the PCM-16x0 analysis, the prescans and the audio plan
bash $R/tools/gpu_pmc_round3.sh 2>&1 | grep "rc="
# the damaged-tape kernels (general frame kernel, the two sweep kernels, the histogram carry) on the C3 PAL tape
bash $R/tools/gpu_pmc_round4.sh 2>&1 | grep "rc="
# the summaries are made here (same tree, same source hash) and travel back under gpurun_out/r06_profiles/; the raw counter and trace files stay
# on the box (gpurun only merges 64 MiB back)
cd $R && python tools/refresh_profiles.py r06 > gpurun_out/refresh_profiles.log 2>&1; echo "refresh rc=$?"
rm -rf gpurun_out/r06_profiles && mkdir -p gpurun_out/r06_profiles && cp profiles/r06_* gpurun_out/r06_profiles/ && rm -f gpurun_out/r06_profiles/*.md gpurun_out/r06_profiles/*.txt   # (the hand-written notes and tables do not come back: a copy from the box would overwrite what was written meanwhile)
find gpurun_out -name "*.csv" -size +256k -delete; find gpurun_out -name "*.db" -delete
du -sh gpurun_out | tail -1
# scheduler traces of the two damaged tapes (developer build, when one was sent along)
if [ -f build/variants/dev.so ]; then
  SDVPCM_LIB=build/variants/dev.so SDV_SCHED_TRACE=1 timeout 300 python tools/pal_trace.py 2000 both > gpurun_out/pal_trace_both.log 2>&1
  SDVPCM_LIB=build/variants/dev.so SDV_SCHED_TRACE=1 timeout 300 python tools/jump_probe.py 10000 16 > gpurun_out/jump_trace.log 2>&1
fi
timeout 1200 python tools/soak.py 12 3000 > gpurun_out/soak_r06.log 2>&1; echo "soak rc=$?"; tail -3 gpurun_out/soak_r06.log
