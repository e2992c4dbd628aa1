#!/usr/bin/env python3
"""Build container: averages per launch of the counters in gpurun_out/pmc_<tag>_*/ for kernels matching a substring.
usage: pmc_sum.py <tag> <kernel substring> [out.json]"""
import collections, csv, glob, json, sys
tag, sub = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("gpurun_out/pmc_%s_*/**/*counter_collection.csv" % tag, recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if sub in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}
for k, d in out.items():
    print(k)
    for c, v in d.items():
        print("   %-34s %.5g" % (c, v))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
