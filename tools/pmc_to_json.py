#!/usr/bin/env python3
"""gpurun_out/pmc1..pmc4 (rocprofv3 --pmc passes of tools/gpu_validate.sh) -> profiles/<round>_pmc_<kernel>.json:
per-launch counter sums of the last full-batch dispatch of the given kernel."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sdvpcmdecoder_amd.build import source_hash
kernel = sys.argv[1] if len(sys.argv) > 1 else "sdv_k_stc007_frames_lean"
out = sys.argv[2] if len(sys.argv) > 2 else "profiles/r06_pmc_%s.json" % kernel
prefix = sys.argv[3] if len(sys.argv) > 3 else "pmc"          # directory prefix under gpurun_out/ (pmc1..4, p1pmc1..4)
res = {}
for p in tuple(prefix + str(i) for i in (1, 2, 3, 4)):
    agg = collections.OrderedDict()
    files = glob.glob("gpurun_out/%s/**/*_counter_collection.csv" % p, recursive=True)
    files.sort(key=os.path.getmtime)
    for f in files[-1:]:            # gpurun merges into gpurun_out/: older runs' files stay around - only the newest run counts
        for r in csv.DictReader(open(f)):
            if not r["Kernel_Name"].startswith(kernel + "("):
                continue
            k = (int(r["Dispatch_Id"]), int(r["Grid_Size"]))
            d = agg.setdefault(k, {"vgpr": r["VGPR_Count"], "sgpr": r["SGPR_Count"], "scratch": r["Scratch_Size"], "lds": r["LDS_Block_Size"]})
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    if not agg:
        continue
    big = max(g for (_, g) in agg)
    last = max(k for k in agg if k[1] == big)
    res[p] = dict(agg[last], grid_size=big)
# what was measured: the sources of the library (run this right after the passes, on the tree that was sent to the GPU box) and the
# workload of the profiled command (argv[4], e.g. "frames=10000,mode=2,noise=4.0,width=720,height=486") - bench.py quotes the
# counters only for the same build and workload
res["source_sha16"] = source_hash(kernel)
res["workload"] = sys.argv[4] if len(sys.argv) > 4 else "frames=10000,mode=2,noise=4.0,width=720,height=486"
json.dump(res, open(out, "w"), indent=1)
print(out, {p: {k: v for k, v in d.items() if k.isupper()} for p, d in res.items() if isinstance(d, dict)})
