"""FETCH_SIZE / WRITE_SIZE (or any counters) of one kernel from rocprofv3 --pmc output directories: per-launch sums of the full-batch dispatches.
usage: pmc_sizes.py kernel dir [dir ...]"""
import collections, csv, glob, os, sys
kernel = sys.argv[1]
for d in sys.argv[2:]:
    agg = collections.OrderedDict()
    for f in sorted(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(f)):
            if not r["Kernel_Name"].startswith(kernel + "("):
                continue
            k = (int(r["Dispatch_Id"]), int(r["Grid_Size"]))
            agg.setdefault(k, {}).setdefault(r["Counter_Name"], 0.0)
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if not agg:
        print(d, "no dispatches of", kernel); continue
    big = max(g for (_, g) in agg)
    rows = [v for k, v in agg.items() if k[1] == big]
    names = sorted(rows[-1])
    print(d, "grid", big, "launches", len(rows), " ".join("%s=%.0f" % (nm, rows[-1][nm]) for nm in names))
