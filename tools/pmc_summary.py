#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv per dispatch (dev tool)."""
import csv, glob, collections, sys
d = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/pmc1'
for f in glob.glob(d + '/**/*_counter_collection.csv', recursive=True):
    agg = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = (r['Dispatch_Id'], r['Grid_Size'], r['Kernel_Name'][:30], r['VGPR_Count'], r['SGPR_Count'], r['Scratch_Size'])
        agg.setdefault(k, {})[r['Counter_Name']] = agg.setdefault(k, {}).get(r['Counter_Name'], 0) + float(r['Counter_Value'])
    for k, v in agg.items():
        waves = v.get('SQ_WAVES', 1) or 1
        print(k, {n: round(x / waves) for n, x in v.items()})
