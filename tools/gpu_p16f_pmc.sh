# PMC passes of the PCM-1 / PCM-16x0 frame kernels:  gpurun -- 'bash tools/gpu_p16f_pmc.sh'
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for fmt in pcm16 pcm1; do
  rm -rf $R/gpurun_out/${fmt}f_pmc1 $R/gpurun_out/${fmt}f_pmc2
  timeout 200 rocprofv3 --kernel-include-regex "sdv_k_${fmt}_frames_bin" --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $R/gpurun_out/${fmt}f_pmc1 -- python3 $R/tools/${fmt}_frames_prof.py 2000 1 > /dev/null 2> $R/gpurun_out/${fmt}f_pmc1.err; echo "${fmt} pmc1 rc=$?"
  timeout 200 rocprofv3 --kernel-include-regex "sdv_k_${fmt}_frames_bin" --pmc SQ_WAVES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM --output-format csv -d $R/gpurun_out/${fmt}f_pmc2 -- python3 $R/tools/${fmt}_frames_prof.py 2000 1 > /dev/null 2> $R/gpurun_out/${fmt}f_pmc2.err; echo "${fmt} pmc2 rc=$?"
done
cd $R
python3 - <<'PY'
import csv, glob, collections
for fmt in ('pcm16', 'pcm1'):
    for d in (fmt + 'f_pmc1', fmt + 'f_pmc2'):
        fs = sorted(glob.glob(f'gpurun_out/{d}/**/*counter_collection.csv', recursive=True))
        if not fs: print(d, 'no file'); continue
        agg = collections.OrderedDict()
        for r in csv.DictReader(open(fs[-1])):
            k = int(r['Dispatch_Id'])
            a = agg.setdefault(k, {'grid': int(r['Grid_Size']), 'vgpr': r['VGPR_Count'], 'scratch': r['Scratch_Size'], 'lds': r['LDS_Block_Size']})
            a[r['Counter_Name']] = a.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
        for k, a in agg.items():
            n = a['grid'] / 64
            print(d, 'dispatch', k, 'frames', int(n), {c: round(v / n) for c, v in a.items() if c.isupper()}, 'vgpr', a['vgpr'], 'scratch', a['scratch'], 'lds', a['lds'])
PY
