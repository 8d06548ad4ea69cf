"""Summarises rocprofv3 --pmc counter_collection.csv files: mean per kernel and counter.
usage: python tools/pmc_summary.py gpurun_out/<dir> ..."""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    for f in sorted(glob.glob(d + "/*/*counter_collection.csv")):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in sorted(agg.items()):
            if "mtr::" in k:
                print(f"{d.split('/')[-1]},{k.split('(')[0]}," + ",".join(f"{c}={sum(x)/len(x):.1f}" for c, x in sorted(v.items())) + f",n={len(next(iter(v.values())))}")
