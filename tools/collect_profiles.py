#!/usr/bin/env python3
"""Aggregates the rocprofv3 outputs of a profiling call (gpurun_out/<dirs>) into profiles/<round>_*.
Usage: collect_profiles.py <round> <stats_dir> <fetch_dir> <write_dir> <sq_dir> <bench_profiled.json> <bench_default.json>

The profiling call (one gpurun, from the repository root; `cd /tmp && export TMPDIR=/tmp` first is not needed when -d is given):
  python3 bench.py > gpurun_out/bench_default.json
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_stats -- python3 bench.py --no-cpu > gpurun_out/bench_profiled.json
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --pairs 32768 --steps 1 --warmup 0 --no-cpu
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --pairs 32768 --steps 1 --warmup 0 --no-cpu
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace
            --output-format csv -d gpurun_out/pmc_sq -- python3 bench.py --pairs 32768 --steps 1 --warmup 0 --no-cpu
"""
import collections, csv, glob, json, re, shutil, sys

# units one launch of the kernel processes in `bench.py --pairs 32768` (pairs, scans, poses)
UNITS = {"lsa::icp_kernel": 32768, "lsa::psm_kernel": 32768, "lsa::plicp_kernel": 8192, "lsa::preprocess_lanes_kernel": 16384,
         "lsa::grid_binned_kernel": 2048, "lsa::grid_prep_kernel": 2048, "lsa::score_poses_kernel": 1 << 20, "lsa::grid_prob_kernel": 1}


def norm(n):
    # template arguments dropped for the benchmark sensor's instances only: bench.py's entries on the reference's logs launch
    # the 181- / 361-beam instances with other pair counts, which stay under their own names (and out of UNITS)
    n = n.split("(")[0].replace("void ", "")
    if re.search(r"::(icp|psm)_kernel<(?!1081>)", n):
        return n
    return re.sub(r"<.*>", "", n)


rnd, stats, fetch, write, sq, bprof, bdef = sys.argv[1:8]
out = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, `python3 bench.py --pairs 32768 --steps 1 "
                "--warmup 0 --no-cpu` (32768 pairs per matcher launch, 2048 scans per grid launch). Values are raw counter KB per "
                "launch, averaged over the launches of the kernel (the warm-up launches of bench.py's timing helper included); "
                "MI355X_MICROARCH.md: on gfx950 FETCH_SIZE under-reports wide streaming reads by 2x (bench.py doubles it), "
                "WRITE_SIZE is exact. icp_kernel WRITE is mostly scratch (register spill) traffic: the algorithmic output is 12 B per pair."}
for d, ctr in ((fetch, "FETCH_SIZE"), (write, "WRITE_SIZE")):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = norm(r["Kernel_Name"])
        if r["Counter_Name"] == ctr and k.startswith("lsa::"):
            acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
    for k, (n, v) in acc.items():
        out.setdefault(k, {})[ctr + "_KB_per_launch"] = v / n
for k, u in UNITS.items():
    if k in out:
        out[k]["units_per_launch"] = u
json.dump(out, open("profiles/%s_pmc_hbm_traffic.json" % rnd, "w"), indent=1)
f = glob.glob(sq + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(f)):
    k = norm(r["Kernel_Name"])
    if k.startswith("lsa::"):
        e = acc[k][r["Counter_Name"]]; e[0] += 1; e[1] += float(r["Counter_Value"])
sqo = {k: {c: v[1] / v[0] for c, v in d.items()} for k, d in acc.items()}
for k, u in UNITS.items():
    if k in sqo:
        sqo[k]["units_per_launch"] = u
sqo["_note"] = ("rocprofv3 --pmc SQ_* GRBM_GUI_ACTIVE, `python3 bench.py --pairs 32768 --steps 1 --warmup 0 --no-cpu`, averages per launch "
                "(32768 pairs per matcher launch). SQ cycle counters are in quad-cycles (MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is "
                "summed over the 8 XCDs.")
json.dump(sqo, open("profiles/%s_pmc_sq_counters.json" % rnd, "w"), indent=1)
shutil.copy(glob.glob(stats + "/**/*kernel_stats.csv", recursive=True)[0], "profiles/%s_kernel_stats.csv" % rnd)
shutil.copy(bprof, "profiles/%s_bench_profiled.json" % rnd)
shutil.copy(bdef, "profiles/%s_bench_default.json" % rnd)
for k in ("lsa::icp_kernel", "lsa::psm_kernel", "lsa::plicp_kernel", "lsa::grid_binned_kernel"):
    print(k, out.get(k), {c: "%.4g" % v for c, v in sqo.get(k, {}).items()})
