# instruction mix and waits of one kernel under one build: gpurun -- 'bash tools/gpu_pmc_quick.sh build/variants/x.so "regex" tools/some_prof.py args'
R=$GRAFT_REPO_ROOT
export SDVPCM_LIB=$R/$1; K=$2; shift 2
tool=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/q1 $R/gpurun_out/q2
timeout 600 rocprofv3 --kernel-include-regex "$K" --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/q1 -- python3 $tool "$@" > /dev/null 2> $R/gpurun_out/q1.err; echo "q1 rc=$?"
timeout 600 rocprofv3 --kernel-include-regex "$K" --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_FLAT --output-format csv -d $R/gpurun_out/q2 -- python3 $tool "$@" > /dev/null 2> $R/gpurun_out/q2.err; echo "q2 rc=$?"
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ['GRAFT_REPO_ROOT']
for d in ('q1', 'q2'):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for f in glob.glob(R + '/gpurun_out/' + d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0]
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
            n[(k, r['Counter_Name'])] += 1
    for k, v in acc.items():
        print(k, {c: round(x / max(1, n[(k, c)])) for c, x in v.items()}, 'dispatches', max(n[(k, c)] for c in v))
PY
rm -rf $R/gpurun_out/q1 $R/gpurun_out/q2
