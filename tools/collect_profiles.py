#!/usr/bin/env python3
"""Aggregates the rocprofv3 outputs of a profiling call (gpurun_out/<dirs>) into profiles/r01_*.
Usage: collect_profiles.py <stats_dir> <fetch_dir> <write_dir> <sq_dir> <bench_profiled.json> <bench_default.json>"""
import collections, csv, glob, json, re, shutil, sys


def norm(n):
    n = n.split("(")[0].replace("void ", "")
    return re.sub(r"<.*>", "", n)


stats, fetch, write, sq, bprof, bdef = sys.argv[1:7]
out = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, `python3 bench.py --pairs 32768 --steps 1 "
                "--warmup 0 --no-cpu --extras` (32768 pairs per matcher launch, 2048 scans per grid launch). Values are raw counter KB per "
                "launch; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE under-reports wide streaming reads by 2x (bench.py doubles it), "
                "WRITE_SIZE is exact. Calibration on preprocess_kernel (known 6.92 MB read, 12.1 MB written): raw FETCH 4.5 MB "
                "(x1.5), WRITE 12.2 MB (x1.00). icp_kernel WRITE is scratch (register spill) traffic: the algorithmic output is "
                "12 B per pair."}
for d, ctr in ((fetch, "FETCH_SIZE"), (write, "WRITE_SIZE")):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = norm(r["Kernel_Name"])
        if r["Counter_Name"] == ctr and k.startswith("lsa::"):
            acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    for k, (n, v) in acc.items():
        out.setdefault(k, {})[ctr + "_KB_per_launch"] = v / n
for k in ("lsa::icp_kernel", "lsa::psm_kernel"):
    out[k]["pairs_per_launch"] = 32768
json.dump(out, open("profiles/r01_pmc_hbm_traffic.json", "w"), indent=1)
f = glob.glob(sq + "/*/*counter_collection.csv")[0]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(f)):
    k = norm(r["Kernel_Name"])
    if k.startswith("lsa::"):
        e = acc[k][r["Counter_Name"]]; e[0] += 1; e[1] += float(r["Counter_Value"])
sqo = {k: {c: v[1] / v[0] for c, v in d.items()} for k, d in acc.items()}
sqo["_note"] = ("rocprofv3 --pmc SQ_* GRBM_GUI_ACTIVE, `python3 bench.py --pairs 32768 --steps 1 --warmup 0 --no-cpu --extras`, averages per launch "
                "(32768 pairs per matcher launch). SQ cycle counters are in quad-cycles (MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is "
                "summed over the 8 XCDs.")
json.dump(sqo, open("profiles/r01_pmc_sq_counters.json", "w"), indent=1)
shutil.copy(glob.glob(stats + "/*/*kernel_stats.csv")[0], "profiles/r01_kernel_stats.csv")
shutil.copy(bprof, "profiles/r01_bench_profiled.json")
shutil.copy(bdef, "profiles/r01_bench_default.json")
for k in ("lsa::icp_kernel", "lsa::psm_kernel", "lsa::grid_binned_kernel"):
    print(k, out.get(k), {c: "%.4g" % v for c, v in sqo.get(k, {}).items()})
