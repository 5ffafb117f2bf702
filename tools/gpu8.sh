cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err; echo "rc=$?"; cat gpurun_out/bench_full.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_bench.err; echo "rc=$?"
cat $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json
rocprofv3 --kernel-include-regex 'sdv_k_stc007' --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/pmc1.err; echo "rc=$?"
rocprofv3 --kernel-include-regex 'sdv_k_stc007' --pmc SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/pmc2.err; echo "rc=$?"
rocprofv3 --kernel-include-regex 'sdv_k_stc007' --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/pmc3.err; echo "rc=$?"
rocprofv3 --kernel-include-regex 'sdv_k_stc007' --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc4 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/pmc4.err; echo "rc=$?"
