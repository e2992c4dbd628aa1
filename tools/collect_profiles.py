#!/usr/bin/env python3
"""Build container: turns the outputs of tools/profile_round.sh (gpurun_out/<round>/) into profiles/<round>_*.
Usage: collect_profiles.py <round>

Per kernel = per full name, template arguments kept (`lsa::icp_kernel<361, 128>`, `lsa::plicp_kernel<3, false>`): the sensor
sizes run different instances.  Launches of fewer than 64 workgroups are left out of the counter averages (the single-pair
calls of bench.py's latency entries would otherwise sit in the same rows as the batches).  `units_per_launch` is what one
remaining launch of `bench.py --pairs 32768 --steps 1 --warmup 0 --no-cpu` processes (pairs, scans, poses); kernels launched
with several sizes under one name have none - except p2p_icp_kernel, whose unit is the workgroup (one problem each)."""
import collections, csv, glob, json, os, re, shutil, sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = "gpurun_out/%s" % rnd
UNITS = {"lsa::icp_kernel<1081, 256>": 32768, "lsa::psm_kernel<1081, 256>": 32768, "lsa::plicp_kernel<256, 3, false>": 32768,
         "lsa::icp_kernel<361, 128>": 131072, "lsa::psm_kernel<361, 64>": 131072, "lsa::icp_kernel<181, 64>": 131072, "lsa::psm_kernel<181, 64>": 131072,
         "lsa::icp_kernel<541, 128>": 131072, "lsa::psm_kernel<541, 128>": 131072, "lsa::plicp_kernel<64, 4, false>": 131072, "lsa::plicp_kernel<64, 3, false>": 131072, "lsa::plicp_kernel<128, 4, false>": 131072, "lsa::plicp_kernel<128, 3, false>": 131072,
         "lsa::preprocess_lanes_kernel": 16384, "lsa::grid_binned_kernel": 2048, "lsa::grid_prep_kernel": 2048, "lsa::grid_prob_kernel": 1}


def norm(n):
    return re.sub(r"\(.*$", "", n.replace("void ", "")).strip()


def read(d):
    """{kernel: {counter: [mean per launch, launches, mean workgroups]}} over the launches of >= 64 workgroups"""
    fs = sorted(glob.glob("%s/%s/**/*counter_collection.csv" % (src, d), recursive=True), key=os.path.getmtime, reverse=True)   # (the newest run)
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0, 0.0]))
    if not fs:
        return {}
    for r in csv.DictReader(open(fs[0])):
        k = norm(r["Kernel_Name"])
        wgs = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
        if not k.startswith("lsa::") or wgs < 64:
            continue
        e = acc[k][r["Counter_Name"]]
        e[0] += float(r["Counter_Value"]); e[1] += 1; e[2] += wgs
    return {k: {c: (v[0] / v[1], v[1], v[2] / v[1]) for c, v in d.items()} for k, d in acc.items()}


hbm = {"_note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes of `python3 bench.py --pairs 32768 --steps 1 --warmup 0 "
                "--no-cpu` (tools/profile_round.sh).  Raw counter KB per launch, averaged over the kernel's launches of >= 64 workgroups "
                "(the warm-up launches of bench.py's timing helper included).  MI355X_MICROARCH.md: on gfx950 FETCH_SIZE under-reports "
                "wide streaming reads by 2x (bench.py doubles it), WRITE_SIZE is exact."}
for d, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    for k, cs in read(d).items():
        if ctr in cs:
            hbm.setdefault(k, {})[ctr + "_KB_per_launch"] = cs[ctr][0]
            hbm[k]["launches"] = cs[ctr][1]
            if k.startswith("lsa::p2p_icp_kernel"):
                hbm[k]["units_per_launch"] = cs[ctr][2]
for k, u in UNITS.items():
    if k in hbm:
        hbm[k]["units_per_launch"] = u
for k, e in hbm.items():
    if isinstance(e, dict) and e.get("units_per_launch") and "WRITE_SIZE_KB_per_launch" in e:
        e["WRITE_bytes_per_unit"] = e["WRITE_SIZE_KB_per_launch"] * 1024.0 / e["units_per_launch"]
        if "FETCH_SIZE_KB_per_launch" in e:
            e["FETCH_bytes_per_unit_x2"] = 2.0 * e["FETCH_SIZE_KB_per_launch"] * 1024.0 / e["units_per_launch"]
json.dump(hbm, open("profiles/%s_pmc_hbm_traffic.json" % rnd, "w"), indent=1)

sq = {}
for k, cs in read("pmc_sq").items():
    sq[k] = {c: v[0] for c, v in cs.items()}
    sq[k]["launches"] = next(iter(cs.values()))[1]
    if k.startswith("lsa::p2p_icp_kernel"):
        sq[k]["units_per_launch"] = next(iter(cs.values()))[2]
for k, u in UNITS.items():
    if k in sq:
        sq[k]["units_per_launch"] = u
for k, e in sq.items():
    if e.get("SQ_WAVE_CYCLES"):
        e["wave_wait_frac"] = e.get("SQ_WAIT_ANY", 0.0) / e["SQ_WAVE_CYCLES"]
    if e.get("SQ_BUSY_CU_CYCLES"):
        e["valu_busy_frac"] = 2.0 * e.get("SQ_ACTIVE_INST_VALU", 0.0) / (4.0 * e["SQ_BUSY_CU_CYCLES"])
sq["_note"] = ("rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CU_CYCLES "
               "GRBM_GUI_ACTIVE, one pass of `python3 bench.py --pairs 32768 --steps 1 --warmup 0 --no-cpu`; averages per launch over the launches "
               "of >= 64 workgroups.  wave_wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES; valu_busy_frac = 2 cycles x SQ_ACTIVE_INST_VALU / (4 SIMDs x "
               "SQ_BUSY_CU_CYCLES): the counter counts issue units (1.01 per wave64 instruction in pm_icp), a full-rate unit occupies a "
               "SIMD-32 for two cycles.  GRBM_GUI_ACTIVE is summed over the 8 XCDs.")
json.dump(sq, open("profiles/%s_pmc_sq_counters.json" % rnd, "w"), indent=1)

# the pose scorer alone (tools/score_probe.py: 2^20 poses x 361 beams per launch)
sc = {"_note": "tools/score_probe.py under rocprofv3 --pmc, one counter group per pass (SQ | TCP | TCP stalls | TCC): lsa::pf_score_kernel, "
               "2^20 poses x 361 beams per launch = 378 535 936 lane-beams; averages per launch.", "units_per_launch": 1 << 20, "beams_per_pose": 361}
for d in ("score_sq", "score_tcp", "score_tcp2", "score_tcc"):
    for k, cs in read(d).items():
        if "pf_score_kernel" in k:
            sc["kernel"] = k
            for c, v in cs.items():
                sc[c] = v[0]
lb = float(sc["units_per_launch"]) * sc["beams_per_pose"]
if "TCP_TOTAL_CACHE_ACCESSES_sum" in sc:
    sc["tcp_accesses_per_lane_beam"] = sc["TCP_TOTAL_CACHE_ACCESSES_sum"] / lb
if "TCP_TCC_READ_REQ_sum" in sc:
    sc["l2_requests_per_lane_beam"] = sc["TCP_TCC_READ_REQ_sum"] / lb
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in sc:
        sc["tcp_hit_rate"] = 1.0 - sc["TCP_TCC_READ_REQ_sum"] / sc["TCP_TOTAL_CACHE_ACCESSES_sum"]
if "TCC_HIT_sum" in sc and "TCC_MISS_sum" in sc:
    sc["tcc_hit_rate"] = sc["TCC_HIT_sum"] / (sc["TCC_HIT_sum"] + sc["TCC_MISS_sum"])
if "SQ_INSTS_VALU" in sc:
    sc["valu_instr_per_lane_beam"] = sc["SQ_INSTS_VALU"] * 64.0 / lb
if sc.get("SQ_WAVE_CYCLES"):
    sc["wave_wait_frac"] = sc.get("SQ_WAIT_ANY", 0.0) / sc["SQ_WAVE_CYCLES"]
    sc["wave_wait_inst_frac"] = sc.get("SQ_WAIT_INST_ANY", 0.0) / sc["SQ_WAVE_CYCLES"]
if sc.get("SQ_BUSY_CU_CYCLES"):
    sc["valu_busy_frac"] = 2.0 * sc.get("SQ_ACTIVE_INST_VALU", 0.0) / (4.0 * sc["SQ_BUSY_CU_CYCLES"])
if sc.get("TCP_GATE_EN1_sum"):
    sc["tcp_pending_stall_frac"] = sc.get("TCP_PENDING_STALL_CYCLES_sum", 0.0) / sc["TCP_GATE_EN1_sum"]
json.dump(sc, open("profiles/%s_pmc_score_poses.json" % rnd, "w"), indent=1)

st = sorted(glob.glob("%s/stats/**/*kernel_stats.csv" % src, recursive=True), key=os.path.getmtime, reverse=True)
if st:
    shutil.copy(st[0], "profiles/%s_kernel_stats.csv" % rnd)
for a, b in (("bench_profiled.json", "bench_profiled.json"), ("bench_default.json", "bench_default.json")):
    try:
        shutil.copy("%s/%s" % (src, a), "profiles/%s_%s" % (rnd, b))
    except OSError:
        pass
for k in ("lsa::icp_kernel<1081, 256>", "lsa::psm_kernel<1081, 256>", "lsa::plicp_kernel<256, 3, false>", "lsa::plicp_kernel<128, 4, false>", "lsa::p2p_icp_kernel<128>", "lsa::icp_kernel<361, 128>", "lsa::psm_kernel<361, 64>", "lsa::icp_kernel<181, 64>"):
    print(k, {c: "%.4g" % v for c, v in hbm.get(k, {}).items()}, {c: "%.4g" % v for c, v in sq.get(k, {}).items()})
print("score:", {c: ("%.4g" % v if isinstance(v, float) else v) for c, v in sc.items() if c != "_note"})
