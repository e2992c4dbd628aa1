#!/usr/bin/env python3
"""Diagnostic: per-kernel averages of the counters in rocprofv3 counter_collection.csv files under the given directories."""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if k.startswith("lsa::"):
                e = acc[k][r["Counter_Name"]]; e[0] += 1; e[1] += float(r["Counter_Value"])
        for k, c in acc.items():
            print(d, k, " ".join("%s=%.4g" % (n, v[1] / v[0]) for n, v in sorted(c.items())))
