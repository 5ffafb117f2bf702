cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in w7 w8 w9 w10 w16; do
  SDVPCM_LIB=$GRAFT_REPO_ROOT/build/variants/lib_$w.so python bench.py --steps 5 --warmup 1 --no-cpu > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err; echo "$w rc=$?"
  python -c "
import json;d=json.load(open('gpurun_out/bench_$w.json'));print('$w', d['value'], d['roofline']['avg_launch_ms'], d['config']['decoded_words_match_generator'])"
done
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc1 $GRAFT_REPO_ROOT/gpurun_out/pmc2
SDVPCM_LIB=$GRAFT_REPO_ROOT/build/variants/lib_w8.so rocprofv3 --kernel-include-regex 'sdv_k_stc007' --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/pmc1.err; echo "rc=$?"
SDVPCM_LIB=$GRAFT_REPO_ROOT/build/variants/lib_w8.so rocprofv3 --kernel-include-regex 'sdv_k_stc007' --pmc SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/pmc2.err; echo "rc=$?"
