# PMC passes + kernel stats of the PCM-1 line kernel (sdv_k_pcm1_lines):  gpurun -- 'bash tools/gpu_pcm1_front_pmc.sh'
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_pcm1f $R/gpurun_out/p1fpmc1 $R/gpurun_out/p1fpmc2 $R/gpurun_out/p1fpmc3 $R/gpurun_out/p1fpmc4
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_pcm1f -- python3 $R/tools/pcm1_front_prof.py 2000 10 > $R/gpurun_out/prof_pcm1f.log 2>&1; echo "rocprof pcm1 front rc=$?"
timeout 600 rocprofv3 --kernel-include-regex 'sdv_k_pcm1_lines' --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/p1fpmc1 -- python3 $R/tools/pcm1_front_prof.py 2000 1 > /dev/null 2> $R/gpurun_out/p1fpmc1.err; echo "p1fpmc1 rc=$?"
timeout 600 rocprofv3 --kernel-include-regex 'sdv_k_pcm1_lines' --pmc SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/p1fpmc2 -- python3 $R/tools/pcm1_front_prof.py 2000 1 > /dev/null 2> $R/gpurun_out/p1fpmc2.err; echo "p1fpmc2 rc=$?"
timeout 600 rocprofv3 --kernel-include-regex 'sdv_k_pcm1_lines' --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/p1fpmc3 -- python3 $R/tools/pcm1_front_prof.py 2000 1 > /dev/null 2> $R/gpurun_out/p1fpmc3.err; echo "p1fpmc3 rc=$?"
timeout 600 rocprofv3 --kernel-include-regex 'sdv_k_pcm1_lines' --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/p1fpmc4 -- python3 $R/tools/pcm1_front_prof.py 2000 1 > /dev/null 2> $R/gpurun_out/p1fpmc4.err; echo "p1fpmc4 rc=$?"
cd $R
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob('gpurun_out/prof_pcm1f/*/*kernel_stats.csv'))[-1]
for r in csv.DictReader(open(f)):
    if r['Name'].startswith('sdv_'):
        print(r['Name'][:44], r['Calls'], 'avg us', round(float(r['AverageNs']) / 1e3, 1), 'min', round(float(r['MinNs']) / 1e3, 1), 'max', round(float(r['MaxNs']) / 1e3, 1))
for d in ('p1fpmc1', 'p1fpmc2', 'p1fpmc3', 'p1fpmc4'):
    fs = sorted(glob.glob(f'gpurun_out/{d}/*/*counter_collection.csv'))
    if not fs: continue
    rows = list(csv.DictReader(open(fs[-1])))
    # the big launch (the warm one): most workgroups
    # the lean kernel's big launch (the warm one): per dispatch sums, the last one
    acc = {}
    for r in rows:
        if r['Kernel_Name'].startswith('sdv_k_pcm1_lines_lean('):
            acc.setdefault(int(r['Dispatch_Id']), {}).setdefault(r['Counter_Name'], 0.0)
            acc[int(r['Dispatch_Id'])][r['Counter_Name']] += float(r['Counter_Value'])
    if acc: print(d, {k: round(v) for k, v in acc[max(acc)].items()})
PY
