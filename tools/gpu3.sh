set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for w in 1 3 4 6; do
  SDVPCM_LIB=$GRAFT_REPO_ROOT/build/variants/lib_w$w.so python bench.py --steps 5 --warmup 1 --no-cpu > gpurun_out/bench_w$w.json 2> gpurun_out/bench_w$w.err; echo "w=$w rc=$?"
  python -c "
import json;d=json.load(open('gpurun_out/bench_w$w.json'));print('w$w', d['value'], d['roofline']['avg_launch_ms'], d['config']['decoded_words_match_generator'])"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_bench.err; echo "rc=$?"
find $GRAFT_REPO_ROOT/gpurun_out/prof_r1 -name '*.csv' | head
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/pmc1.err; echo "rc=$?"
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/pmc2.err; echo "rc=$?"
find $GRAFT_REPO_ROOT/gpurun_out/pmc1 $GRAFT_REPO_ROOT/gpurun_out/pmc2 -name '*.csv' | head
