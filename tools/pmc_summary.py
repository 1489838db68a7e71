#!/usr/bin/env python3
"""tools/pmc_summary.py <dir> [kernel-substring] -- mean per-dispatch PMC values + kernel stats as markdown"""
import collections, csv, glob, sys
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else "lgpu"
print("## rocprofv3 --kernel-trace --stats (lgpu kernels)")
print("| kernel | calls | avg ns | min ns | max ns | % |\n|---|---|---|---|---|---|")
for r in csv.DictReader(open(d + "/trace/t_kernel_stats.csv")):
    if "lgpu" in r["Name"]:
        print("| %s | %s | %.0f | %s | %s | %s |" % (r["Name"].replace("|", "/")[:90], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"], r["Percentage"]))
print("\n## rocprofv3 --pmc (separate passes), mean per dispatch of kernels matching '%s'" % pat)
print("| counter | mean / dispatch | dispatches |\n|---|---|---|")
for f in sorted(glob.glob(d + "/pmc_*/*_counter_collection.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print("| %s | %.5g | %d |" % (k, sum(v) / len(v), len(v)))
