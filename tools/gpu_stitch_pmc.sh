# PMC counters of the stitch turn kernel (instruction mix per turn)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/stpmc1 $R/gpurun_out/stpmc2
rocprofv3 --kernel-include-regex 'sdv_k_stitch_step' --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/stpmc1 -- python3 $R/tools/stitch_prof.py 10000 2 cont > /dev/null 2> $R/gpurun_out/stpmc1.err; echo "rc=$?"
rocprofv3 --kernel-include-regex 'sdv_k_stitch_step' --pmc SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA --output-format csv -d $R/gpurun_out/stpmc2 -- python3 $R/tools/stitch_prof.py 10000 2 cont > /dev/null 2> $R/gpurun_out/stpmc2.err; echo "rc=$?"
cd $R
python3 - <<'PY'
import csv, glob, collections
for d in ('stpmc1', 'stpmc2'):
    f = sorted(glob.glob('gpurun_out/%s/**/*_counter_collection.csv' % d, recursive=True))[-1]
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        agg.setdefault(r['Dispatch_Id'], {})[r['Counter_Name']] = agg.setdefault(r['Dispatch_Id'], {}).get(r['Counter_Name'], 0) + float(r['Counter_Value'])
    k = list(agg)[-1]
    print(d, {n: round(v / 10000) for n, v in agg[k].items()}, '(per turn, 10 000 turns)')
PY
