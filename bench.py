#!/usr/bin/env python3
"""Headline benchmark: scan-matches/s on synthetic 1081-beam scans (BASELINE.json).

A "step" = one batch of `--pairs` scan-pair matches (pm_icp, the matcher behind the reference's FMatch front door)
over a device-resident, already preprocessed scan set.  Inputs are in HBM before the timed region.

`--gpus N`: one process per GPU.  Under torchrun (WORLD_SIZE set) this process is one rank; started plainly with
N > 1 it launches the N ranks itself (child processes, before anything touches the GPU) and rank 0 prints the line.
Each rank matches its own batch (independent pairs -> no data-path collective, weak scaling); value = all pairs of all
ranks / max-over-ranks time.  The particle-filter step, the one part of the path with an exchange, is timed as a
secondary entry over `torch.distributed` (RCCL) with 2^20 particles per rank.

Prints ONE JSON line (rank 0): the contract's fields, `roofline` for the dominant kernel (icp_kernel is bound by
vector-instruction issue: instructions per pair from the committed PMC pass x pairs/s against 256 CUs x 4 SIMDs x
2.4 GHz / 2; the HBM figure of SURVEY.md 8d beside it), `secondary` (the other kernels of the path, each with its own
roofline) and `cpu_baseline` (the reference built in place when oracle/_ref travelled, else the oracle port: single
thread and all host cores, bounded samples, timed before the GPU is touched).
"""
import argparse
import json
import threading
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_PAIR = 2 * 1081 * 4 + 12 + 20      # 8 680 B per 1081-beam pair
HBM_PEAK_GBS = 8000.0                        # MI355X_MICROARCH.md: 8.0 TB/s spec
L2_PEAK_GBS = 34500.0                        # MI355X_MICROARCH.md: 4 MiB per XCD, ~34.5 TB/s aggregate
# wave64 vector instructions/s: 256 CUs x 4 SIMD-32 units, a single-precision instruction every 2 cycles when at least two
# waves of a SIMD are ready (MI355X_MICROARCH.md: `v_fma_f32` 2 cyc, one wave alone 4; transcendentals 8), at 2.4 GHz
VALU_PEAK_GINSTR = 256 * 4 * 2.4 / 2.0
FP32_PEAK_TFLOPS = 157.3
PROFILE_ROUND = "r03"
N_BASE, SEEDS = 200, 8


def kernel_name(matcher, beams):
    """rocprofv3's name of the matcher's throughput instance (template arguments: beams[, threads of the workgroup])."""
    if matcher == "icp":
        return "icp_kernel<%d, %d>" % (beams, {181: 64, 361: 128, 541: 128}.get(beams, 256))
    return "psm_kernel<%d, %d>" % (beams, {181: 64, 361: 64, 541: 128}.get(beams, 256))


def build_scan_set():
    from laser_slam_amd import synth
    sets = [synth.make_scans(N_BASE, noise_seed=synth.WORLD_SEED + j)[0] for j in range(SEEDS)]
    return np.concatenate(sets, 0)


# ---------------------------------------------------------------------------------------------------------------
# CPU baseline: worker processes (started before this process touches the GPU), one matcher object per process
def cpu_worker(raw_path, start, count, which, out_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from laser_slam_amd import synth
    raw = np.load(raw_path)
    use_ref = oracle_lib.have_ref()
    eng = oracle_lib.Ref(**synth.SYN1081) if use_ref else oracle_lib.Oracle(**synth.SYN1081)
    pre = [eng.preprocess(raw[k]) for k in range(start, start + count + 1)]
    pose = np.zeros((count, 3), np.float32); st = np.zeros(count, np.int32)
    t0 = time.perf_counter()
    for b in range(count):
        res = eng.match(which, pre[b], pre[b + 1])
        st[b] = res[0]; pose[b] = res[1]
    dt = time.perf_counter() - t0
    np.savez(out_path, dt=dt, pose=pose, st=st, ref=int(use_ref))


def host_cores():
    """Cores this job may use: the affinity mask, cut to the cgroup's CPU quota where one is set (a GPU box hands a
    one-GPU job a share of its cores; starting a process per visible core would only queue them up)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = txt[0], float(txt[1])
            else:
                quota, period = txt[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                cores = max(1, min(cores, int(float(quota) / period + 0.5)))
            break
        except Exception:
            continue
    cap = os.environ.get("LSA_BENCH_CORES")
    if cap:
        cores = max(1, min(cores, int(cap)))
    return min(cores, 64)


def cpu_baseline(raw, n_pairs, which):
    """Single thread, then all host cores (one process each over disjoint pair ranges, as BASELINE.md 2 timed the
    reference), on bounded samples of the bench workload; preprocessing excluded from the timer."""
    cores = host_cores()
    tmp = tempfile.mkdtemp(prefix="lsa_bench_")
    raw_path = os.path.join(tmp, "raw.npy")
    np.save(raw_path, raw)

    def run(jobs, tag):
        procs = []
        for k, (start, count) in enumerate(jobs):
            out = os.path.join(tmp, "%s%d.npz" % (tag, k))
            procs.append((out, subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", raw_path, str(start),
                                                 str(count), str(which), out], stderr=subprocess.DEVNULL)))
        res = []
        for out, p in procs:
            if p.wait() != 0:
                raise RuntimeError("CPU baseline worker failed")
            with np.load(out) as z:
                res.append({k: z[k] for k in z.files})
        return res

    one = run([(0, n_pairs)], "one")[0]
    per = n_pairs
    jobs = [((k * 97) % (raw.shape[0] - per - 1), per) for k in range(cores)]     # disjoint-ish ranges of the same scan set
    many = run(jobs, "all")
    wall = max(float(r["dt"]) for r in many)
    kind = "reference" if int(one["ref"]) else "port"
    out = {"value": cores * per / wall, "unit": "scan-matches/s", "cores": cores, "kind": kind,
           "sample": "%d processes x %d consecutive synthetic 1081-beam pairs each, pm_icp, one matcher object per process, "
                     "preprocessing excluded (%.1f s)" % (cores, per, wall),
           "single_thread": {"value": n_pairs / float(one["dt"]), "unit": "scan-matches/s", "cores": 1, "kind": kind,
                             "sample": "%d consecutive synthetic 1081-beam pairs, pm_icp, preprocessing excluded (%.1f s)"
                                       % (n_pairs, float(one["dt"]))}}
    return out, one["pose"], one["st"]


# ---------------------------------------------------------------------------------------------------------------
def spawn_ranks(n, argv):
    """--gpus N without torchrun: start the N ranks as child processes (this process never touches the GPU)."""
    import torch
    backend = os.environ.get("LSA_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()                      # (counting devices does not initialise the GPU)
    if backend == "nccl" and ndev < n and "--rehearse-launch" not in argv:
        sys.stderr.write("bench.py: --gpus %d needs %d devices, %d visible (LSA_DIST_BACKEND=gloo lets ranks share devices "
                         "for a rehearsal)\n" % (n, n, ndev))
        return 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    rc = 0
    try:
        for p in procs:
            rc = max(rc, abs(p.wait()))
            if rc:
                break
    finally:
        for p in procs:                                   # a failed rank must not leave the others at a barrier
            if p.poll() is None:
                p.kill()
    return rc


def rehearse_launch(rank, world, args):
    """The N-rank launch path without a GPU: rendezvous, barrier, max-over-ranks reduction, one line from rank 0."""
    import torch
    t = torch.tensor([0.001 * (rank + 1)], dtype=torch.float64)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo")
        dist.barrier()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "scan-matches/sec (1081-beam)", "rehearsal": True, "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "max_over_ranks_s": float(t.item())}), flush=True)
    return 0


def load_profile(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "%s_%s.json" % (PROFILE_ROUND, name))))
    except Exception:
        return None


def pmc_per_unit(kernel):
    """Counters of the committed rocprofv3 PMC passes (profiles/r02_pmc_*.json, tools/collect_profiles.py), per
    launch and per unit (pair / scan): VALU wave-instructions, busy share of the issue slots at the clock the chip
    held, HBM bytes (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE)."""
    out = {}
    sq, hbm = load_profile("pmc_sq_counters"), load_profile("pmc_hbm_traffic")
    key = "lsa::" + kernel
    if sq and key in sq:
        rec = sq[key]
        units = rec.get("units_per_launch")
        if units:
            out["valu_instr_per_unit"] = rec["SQ_INSTS_VALU"] / units
        # share of the busy CUs' SIMD cycles with a vector instruction in flight.  SQ_ACTIVE_INST_VALU counts issue units, not
        # clock cycles: 1.01 per wave64 instruction for pm_icp's mix, where a full-rate instruction occupies a SIMD-32 for two
        # cycles (more for double precision and transcendentals, which count more units) - priced at four cycles, as round 2
        # did, the units exceed the SIMD cycles there are (the 1.077 of its line).  Denominator: SQ_BUSY_CU_CYCLES, in clock
        # cycles per CU, x 4 SIMDs.
        if rec.get("SQ_BUSY_CU_CYCLES"):
            out["valu_busy_frac"] = min(1.0, 2.0 * rec["SQ_ACTIVE_INST_VALU"] / (4.0 * rec["SQ_BUSY_CU_CYCLES"]))
        if rec.get("SQ_WAVE_CYCLES") and rec.get("SQ_WAIT_ANY") is not None:
            out["wave_wait_frac"] = rec["SQ_WAIT_ANY"] / rec["SQ_WAVE_CYCLES"]
        if rec.get("SQ_THREAD_CYCLES_VALU") and rec.get("SQ_INSTS_VALU"):
            out["valu_lane_util"] = rec["SQ_THREAD_CYCLES_VALU"] / (rec["SQ_INSTS_VALU"] * 64.0)
    if hbm and key in hbm:
        rec = hbm[key]
        units = rec.get("units_per_launch") or rec.get("pairs_per_launch")
        if units and "FETCH_SIZE_KB_per_launch" in rec and "WRITE_SIZE_KB_per_launch" in rec:
            out["hbm_bytes_per_unit"] = (2.0 * rec["FETCH_SIZE_KB_per_launch"] + rec["WRITE_SIZE_KB_per_launch"]) * 1024.0 / units
    return out


def valu_roofline(kernel, units_per_s, kernel_ms, units_per_launch, alg_bytes_per_unit, extra=None):
    """Roofline of an instruction-issue-bound matcher kernel: VALU wave-instructions/s against the issue peak, the
    HBM figure (algorithmic bytes of SURVEY.md 8d) beside it."""
    pm = pmc_per_unit(kernel)
    ipu = pm.get("valu_instr_per_unit")
    hbm_gbs = alg_bytes_per_unit * units_per_launch / (kernel_ms * 1e-3) / 1e9
    r = {"bound": "valu", "achieved": ipu * units_per_s / 1e9 if ipu else None, "peak": VALU_PEAK_GINSTR, "unit": "Ginstr/s",
         "frac": ipu * units_per_s / 1e9 / VALU_PEAK_GINSTR if ipu else None,
         "traffic": pm["hbm_bytes_per_unit"] * units_per_launch if "hbm_bytes_per_unit" in pm else None,
         "kernel": kernel, "kernel_ms": kernel_ms, "valu_instr_per_unit": ipu, "valu_busy_frac": pm.get("valu_busy_frac"),
         "valu_lane_util": pm.get("valu_lane_util"), "wave_wait_frac": pm.get("wave_wait_frac"),
         "hbm": {"achieved": hbm_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm_gbs / HBM_PEAK_GBS,
                 "bytes_per_unit": alg_bytes_per_unit},
         "note": "vector-instruction-issue bound (DESIGN.md 4.1): achieved = wave64 VALU instructions per unit (rocprofv3 "
                 "SQ_INSTS_VALU, profiles/%s_pmc_sq_counters.json) x units/s measured here; peak = 256 CUs x 4 SIMD-32 x 2.4 GHz / 2 "
                 "cycles per single-precision instruction (the mix here also holds double-precision and transcendental "
                 "instructions that take 4 and 8, and a lone ready wave issues every 4); valu_busy_frac = 2 cycles x SQ_ACTIVE_INST_VALU / "
                 "(4 SIMDs x SQ_BUSY_CU_CYCLES), the share of the busy CUs' SIMD cycles with a vector instruction in flight; "
                 "wave_wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES; valu_lane_util = active lanes per vector instruction / 64 "
                 "(SQ_THREAD_CYCLES_VALU); hbm = algorithmic bytes of SURVEY 8d" % PROFILE_ROUND}
    if extra:
        r.update(extra)
    return r


def hbm_roofline(kernel, alg_bytes, ms, units_per_launch=None, label=None):
    pm = pmc_per_unit(kernel)
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    r = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
         "traffic": pm["hbm_bytes_per_unit"] * units_per_launch if units_per_launch and "hbm_bytes_per_unit" in pm else None,
         "kernel": kernel, "kernel_ms": ms}
    if label:
        r["note"] = label
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=1 << 20, help="scan pairs per step per GPU (SURVEY 8(d) batch sizes: 1 .. 2^20)")
    ap.add_argument("--matcher", default="icp", choices=["icp", "psm"])
    ap.add_argument("--cpu-pairs", type=int, default=400)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary kernels (profiling passes of the headline alone)")
    ap.add_argument("--cpu-worker", nargs=5, metavar=("RAW", "START", "COUNT", "WHICH", "OUT"), help=argparse.SUPPRESS)
    ap.add_argument("--rehearse-launch", action="store_true",
                    help="launcher plumbing only, no GPU: the ranks meet over gloo and rank 0 prints a line with n_gpus (tests/test_dist_cpu.py)")
    args = ap.parse_args()
    if args.cpu_worker:
        raw_path, start, count, which, out = args.cpu_worker
        cpu_worker(raw_path, int(start), int(count), int(which), out)
        return 0

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: the launcher's world size is used\n" % (args.gpus, world))
    if args.rehearse_launch:
        return rehearse_launch(rank, world, args)

    raw = build_scan_set()                                    # [1600][1081]
    which_i = 1 if args.matcher == "icp" else 0
    cpu = cpu_pose = cpu_st = None
    if rank == 0 and world == 1 and not args.no_cpu:          # before the GPU is touched (worker processes are started)
        cpu, cpu_pose, cpu_st = cpu_baseline(raw, args.cpu_pairs, which_i)

    import torch
    from laser_slam_amd import capi, synth

    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # one process per GPU over RCCL ("nccl" is RCCL on ROCm).  LSA_DIST_BACKEND=gloo lets the
        # multi-rank path be rehearsed on a box with fewer GPUs than ranks (ranks then share devices).
        backend = os.environ.get("LSA_DIST_BACKEND", "nccl")
        ndev = max(1, torch.cuda.device_count())
        torch.cuda.set_device(local_rank % ndev)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank % ndev))
        else:
            dist.init_process_group(backend)
    else:
        torch.cuda.set_device(0)
    dev_index = torch.cuda.current_device()
    coll_dev = "cuda" if (dist is not None and dist.get_backend() == "nccl") else "cpu"

    # ---- workload: synthetic 1081-beam scans, consecutive pairs, zero initial guess ----
    S, n = raw.shape
    p = capi.param_make(**synth.SYN1081)
    ctx = capi.Context(p, dev_index)
    # launch on a torch-owned (non-default) stream so that torch.cuda.Event brackets the kernels
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx.set_stream(stream.cuda_stream)
    d_r = torch.from_numpy(raw).cuda()
    d_bad = torch.zeros(S, n, dtype=torch.uint8, device="cuda")
    d_seg = torch.zeros(S, n, dtype=torch.int16, device="cuda")
    ctx.preprocess_batch(S, d_r, d_bad, d_seg)
    # pairs (k-1, k) inside each seed's run, tiled up to the batch size; each rank takes a
    # different rotation so ranks do not match identical pairs
    base_pairs = np.array([(s * N_BASE + k - 1, s * N_BASE + k) for s in range(SEEDS) for k in range(1, N_BASE)], np.int32)
    B = args.pairs
    sel = (np.arange(B) + rank * 7919) % len(base_pairs)
    d_ri = torch.from_numpy(np.ascontiguousarray(base_pairs[sel, 0])).cuda()
    d_ai = torch.from_numpy(np.ascontiguousarray(base_pairs[sel, 1])).cuda()
    pose = torch.zeros(B, 3, device="cuda")
    st = torch.zeros(B, dtype=torch.int32, device="cuda")
    it = torch.zeros(B, dtype=torch.int32, device="cuda")
    which = capi.MATCH_ICP if args.matcher == "icp" else capi.MATCH_PSM

    def step():
        ctx.match_batch(which, B, d_r, d_bad, d_seg, d_ri, d_ai, pose, d_status=st, d_iters=it)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        step()
        ev[k][1].record()
    barrier()
    elapsed = time.perf_counter() - t0
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    mean_iters = float(it.float().mean().item())
    fail_frac = float((st != 0).float().mean().item())
    total_pairs = B * args.steps * world
    value = total_pairs / elapsed

    out = None
    kern = kernel_name(args.matcher, 1081)
    if rank == 0:
        W = p.window
        flops_pair = mean_iters * (n * 2 * W * 6 + 60 * n) if which == capi.MATCH_ICP else \
            mean_iters / 2 * ((2 * W + 1) * n * 3) + mean_iters * 60 * n
        out = {
            "metric": "scan-matches/sec (1081-beam)", "value": value, "unit": "scan-matches/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "synthetic 1081-beam scans (270 deg, W=200), consecutive pairs, zero initial guess, "
                                   "pm_%s batched, one 256-thread workgroup per pair" % args.matcher,
                       "pairs_per_step_per_gpu": B, "beams": n, "matcher": "pm_" + args.matcher,
                       "mean_iterations": mean_iters, "failed_fraction": fail_frac, "parallelism": "pairs sharded x%d" % world},
            "roofline": valu_roofline(kern, B / (kernel_ms * 1e-3), kernel_ms, B, BYTES_PER_PAIR,
                                      {"reference_flops_per_pair": flops_pair,
                                       "effective_fp32_frac": flops_pair * B / (kernel_ms * 1e-3) / (FP32_PEAK_TFLOPS * 1e12)}),
        }
    # ---- secondary kernels of the path -------------------------------------------------------------------------
    # The headline figure stands from here on.  With several ranks the particle-filter entry below runs collectives that
    # this build container could never exercise on more than one GPU: a watchdog on every rank makes sure that a rank
    # that raises or a collective that never returns costs that entry, not the line - after `deadline` seconds rank 0
    # prints what it has and every rank leaves.
    sec = {}
    printed = threading.Lock()

    def emit_and_leave(note):
        """the headline line with the failure marked, then a non-zero exit status: a launcher that only looks at the status
        must not take a hung or failed multi-rank entry for a passing run"""
        if rank == 0 and printed.acquire(blocking=False):
            out["secondary"] = dict(sec, pf_step={"error": note})
            out["secondary_failed"] = True
            print(json.dumps(out), flush=True)
            sys.stdout.flush()
        os._exit(3)

    watchdog = None
    if world > 1 and not args.no_extras:
        watchdog = threading.Timer(float(os.environ.get("LSA_BENCH_SECONDARY_DEADLINE", "240")), emit_and_leave,
                                   ("multi-rank secondary entries did not finish in time",))
        watchdog.daemon = True
        watchdog.start()
    if not args.no_extras:
        if rank == 0:
            sec.update(secondary_matchers(torch, capi, ctx, d_r, d_bad, d_seg, d_ri, d_ai, B, which, raw, n))
        try:
            sec_pf = secondary_pf(torch, capi, ctx, dist, rank, world, raw, p, coll_dev)     # (all ranks: it has collectives)
        except Exception as e:                              # (the other ranks' watchdogs release them from the collective)
            if world == 1:
                raise
            sec_pf = {"pf_step": {"error": "%s: %s" % (type(e).__name__, e)}}
            if rank != 0:
                os._exit(3)
            emit_and_leave(sec_pf["pf_step"]["error"])
        if rank == 0:
            sec.update(sec_pf)
            if world == 1:
                sec.update(secondary_grid(torch, capi, ctx, raw, p))
                sec.update(secondary_logs(torch, capi))
                sec.update(secondary_latency(torch, capi, raw))
                sec.update(secondary_global(torch, capi, ctx, raw, p))
    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        out["secondary"] = sec
        out["secondary_failed"] = False
        if cpu is not None:
            m = len(cpu_st)
            ri = torch.arange(0, m, dtype=torch.int32, device="cuda"); ai = ri + 1
            gp = torch.zeros(m, 3, device="cuda"); gs = torch.zeros(m, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            ctx.match_batch(which, m, d_r, d_bad, d_seg, ri, ai, gp, d_status=gs); ctx.sync()
            g_pose, g_st = gp.cpu().numpy(), gs.cpu().numpy()
            ok = (cpu_st == 0) & (g_st == 0)
            d = (g_pose[ok].astype(np.float64) - cpu_pose[ok].astype(np.float64))
            cpu["pose_rmse_vs_cpu"] = {
                "pairs": int(m), "matched_by_both": int(ok.sum()), "status_mismatches": int(((cpu_st == 0) != (g_st == 0)).sum()),
                "xy_m": float(np.sqrt(np.mean(d[:, 0] ** 2 + d[:, 1] ** 2)) * 0.01) if ok.any() else None,
                "theta_rad": float(np.sqrt(np.mean(d[:, 2] ** 2))) if ok.any() else None,
                "bit_identical": bool(np.array_equal(g_pose[ok].view(np.uint32), cpu_pose[ok].view(np.uint32)))}
            out["cpu_baseline"] = cpu
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0 and printed.acquire(blocking=False):
        print(json.dumps(out), flush=True)


def timed(torch, fn, reps=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def secondary_matchers(torch, capi, ctx, d_r, d_bad, d_seg, d_ri, d_ai, B, which, raw, n):
    """The other matcher of zhpsm, the point-to-line ICP of zhcsm and the batched preprocessing, each on its own."""
    res = {}
    other = capi.MATCH_PSM if which == capi.MATCH_ICP else capi.MATCH_ICP
    oname = "psm" if other == capi.MATCH_PSM else "icp"
    pose = torch.zeros(B, 3, device="cuda")
    ms = timed(torch, lambda: ctx.match_batch(other, B, d_r, d_bad, d_seg, d_ri, d_ai, pose), reps=3)
    res["pm_%s" % oname] = {"value": B / (ms * 1e-3), "unit": "scan-matches/s", "pairs": B,
                            "roofline": valu_roofline(kernel_name(oname, 1081), B / (ms * 1e-3), ms, B, BYTES_PER_PAIR)}
    # point-to-line ICP (CCanonicalMatcher / zhcsm's matcher), raw scans, 2^17 pairs: a pair that restarts from six
    # perturbed poses runs seven loops, and a small batch waits for its slowest pair (8192 pairs took 17.9 ms in round 2,
    # 9 of them the tail of one such pair)
    Bp = min(B, 1 << 17)
    d_raw = torch.from_numpy(raw).cuda()
    d_o8 = torch.zeros(Bp, 8, dtype=torch.float64, device="cuda")
    ri, ai = d_ri[:Bp].contiguous(), d_ai[:Bp].contiguous()
    ms = timed(torch, lambda: capi.plicp_batch(ctx, Bp, d_raw, ri, ai, d_o8), reps=3)
    res["plicp"] = {"value": Bp / (ms * 1e-3), "unit": "scan-matches/s", "pairs": Bp, "mean_iterations": float(d_o8[:, 5].mean().item()),
                    "roofline": valu_roofline("plicp_kernel<256, 3, false>", Bp / (ms * 1e-3), ms, Bp, BYTES_PER_PAIR,
                                              {"note": "double-precision point-to-line ICP (parity unpinned: CSM absent); "
                                                       "instruction-issue and latency bound, DESIGN.md 4.7"})}
    S = raw.shape[0]
    Sp = 262144                                             # batched preprocessing (one lane per scan through LDS tiles from 65 536 scans on; one wave per scan below)
    r2 = torch.from_numpy(raw).cuda().repeat((Sp + S - 1) // S, 1)[:Sp].contiguous(); b2 = torch.zeros(Sp, n, dtype=torch.uint8, device="cuda"); s2 = torch.zeros(Sp, n, dtype=torch.int16, device="cuda")
    ms = timed(torch, lambda: ctx.preprocess_batch(Sp, r2, b2, s2), reps=3)
    # one fresh scan per launch: what a lone pm_preprocessScan (the head of every FMatch call) costs on the device
    one = [(torch.from_numpy(raw[k:k + 1].copy()).cuda(), torch.zeros(1, n, dtype=torch.uint8, device="cuda"), torch.zeros(1, n, dtype=torch.int16, device="cuda")) for k in range(21)]
    ctx.preprocess_batch(1, *one[0]); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for k in range(1, 21):
        ctx.preprocess_batch(1, *one[k])
    e1.record(); torch.cuda.synchronize()
    res["preprocess_one_scan"] = {"value": e0.elapsed_time(e1) * 1e3 / 20, "unit": "us per scan", "higher_is_better": False, "beams": int(n),
                                  "note": "back-to-back launches of one fresh scan each (one wave per scan, the whole wave at work: DESIGN.md 4.3)"}
    res["preprocess"] = {"value": Sp / (ms * 1e-3), "unit": "scans/s", "scans": Sp,
                         "roofline": hbm_roofline("preprocess_tiles_kernel", Sp * n * 12.0, ms, Sp,
                                                  "12 B per beam: ranges read and written, flags read and written, segments written")}
    return res


def p2p_problems(x_cm, y_cm, ok, pairs, count):
    """Concatenated clouds (metres) and offsets of `count` CICPWarpper problems: the given (reference, current) scan pairs
    of a log head, repeated."""
    clouds = [(x_cm[k][ok[k]] * 0.01, y_cm[k][ok[k]] * 0.01) for k in range(len(x_cm))]
    sel = [pairs[i % len(pairs)] for i in range(count)]
    rx = np.concatenate([clouds[a][0] for a, _ in sel]).astype(np.float32); ry = np.concatenate([clouds[a][1] for a, _ in sel]).astype(np.float32)
    cx = np.concatenate([clouds[b][0] for _, b in sel]).astype(np.float32); cy = np.concatenate([clouds[b][1] for _, b in sel]).astype(np.float32)
    ro = np.concatenate([[0], np.cumsum([len(clouds[a][0]) for a, _ in sel])]).astype(np.int32)
    co = np.concatenate([[0], np.cumsum([len(clouds[b][0]) for _, b in sel])]).astype(np.int32)
    return rx, ry, ro, cx, cy, co


def p2p_entry(torch, capi, ctx, prob, count, what):
    rx, ry, ro, cx, cy, co = prob
    d = [torch.from_numpy(a).cuda() for a in (rx, ry, ro, cx, cy, co)]
    init = torch.zeros(count, 3, dtype=torch.float64, device="cuda"); out = torch.zeros(count, 16, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    ms = timed(torch, lambda: capi.p2p_icp_batch(ctx, count, d[0], d[1], d[2], d[3], d[4], d[5], len(rx), len(cx), init, out), reps=3)
    pts = (len(rx) + len(cx)) / count
    return {"value": count / (ms * 1e-3), "unit": "scan-matches/s", "problems": count, "points_per_problem": pts, "what": what,
            "mean_iterations": float(out[:, 5].mean().item()), "mean_goodness": float(out[:, 3].mean().item()),
            "roofline": valu_roofline("p2p_icp_kernel<%d>" % (64 if pts / 2 <= 200 else 128 if pts / 2 <= 512 else 256), count / (ms * 1e-3), ms, count, pts * 8.0 + 24 + 128,
                                      {"note": "CICPWarpper's point-to-point ICP (MRPT icpClassic restated, parity unpinned): "
                                               "nearest neighbour in a slab of the first map sorted along a per-problem oblique axis (LDS), closed-form 2D fit in double; instruction bound"})}


def secondary_logs(torch, capi, pairs=1 << 17):
    """All four matchers on the heads of the reference's own logs (tests/golden/*_head.*: the three CARMEN logs and the
    in-house LMS151 log, consecutive scans, zero prior), each head's pairs repeated to `pairs` per launch (2^14 problems
    for the point-to-point ICP): real scans need more iterations than the synthetic set and have a sixth to a half of
    its beams.  The zhicp entries are BASELINE configs 1 and 2 (CICPWarpper on fr079 / intel-lab)."""
    res = {}
    for name, preset, fname, tag in (("fr079", "LMS511", "fr079_head.log", "ROBOTLASER1"), ("intel", "LMS211", "intel_head.log", "ROBOTLASER1"),
                                     ("mit", "LMS511", "mit_head.log", "ROBOTLASER1"), ("indoor", "LMS151", "indoor_head.txt", "LMS151")):
        path = os.path.join(ROOT, "tests", "golden", fname)
        if not os.path.exists(path):
            continue
        p = capi.param_preset(preset)
        r, x, y, bad, _ = capi.carmen_read(p, path, tag)
        S, n = r.shape
        ctx = capi.Context(p)
        ctx.set_stream(torch.cuda.current_stream().cuda_stream)        # (so that torch's events bracket the kernels)
        d_raw = torch.from_numpy(r).cuda()
        d_r = d_raw.clone(); d_bad = torch.from_numpy(bad.astype(np.uint8)).cuda()
        d_seg = torch.zeros(S, n, dtype=torch.int16, device="cuda")
        torch.cuda.synchronize()
        ctx.preprocess_batch(S, d_r, d_bad, d_seg)
        idx = (np.arange(pairs) % (S - 1)).astype(np.int32)
        d_ri = torch.from_numpy(idx).cuda(); d_ai = d_ri + 1
        pose = torch.zeros(pairs, 3, device="cuda"); st = torch.zeros(pairs, dtype=torch.int32, device="cuda")
        it = torch.zeros(pairs, dtype=torch.int32, device="cuda")
        entry = {"beams": n, "scans": S, "pairs": pairs}
        bytes_pair = 2 * n * 4 + 12 + 20
        for which, nm, kn in ((capi.MATCH_ICP, "pm_icp", kernel_name("icp", n)), (capi.MATCH_PSM, "pm_psm", kernel_name("psm", n))):
            ms = timed(torch, lambda: ctx.match_batch(which, pairs, d_r, d_bad, d_seg, d_ri, d_ai, pose, d_status=st, d_iters=it), reps=3)
            mi = float(it.float().mean().item())
            entry[nm] = {"value": pairs / (ms * 1e-3), "unit": "scan-matches/s", "mean_iterations": mi,
                         "failed_fraction": float((st != 0).float().mean().item()), "beam_iterations_per_s": pairs / (ms * 1e-3) * mi * n,
                         "roofline": valu_roofline(kn, pairs / (ms * 1e-3), ms, pairs, bytes_pair)}
        d_o8 = torch.zeros(pairs, 8, dtype=torch.float64, device="cuda")
        ms = timed(torch, lambda: capi.plicp_batch(ctx, pairs, d_raw, d_ri, d_ai, d_o8), reps=3)
        entry["plicp"] = {"value": pairs / (ms * 1e-3), "unit": "scan-matches/s", "mean_iterations": float(d_o8[:, 5].mean().item()),
                          "valid_fraction": float(d_o8[:, 3].mean().item())}
        if name in ("fr079", "intel"):
            ok = (bad == 0) & (r < float(p.max_range))
            cnt = 1 << 14
            entry["zhicp_p2p"] = p2p_entry(torch, capi, ctx, p2p_problems(x, y, ok, [(k, k + 1) for k in range(S - 1)], cnt), cnt,
                                           "CICPWarpper::ICPMatch on consecutive scans of the head (reader's points, readings at the maximum range dropped), zero prior")
        ctx.close()
        res["logs_" + name] = entry
    # the reference's own map-node clouds (src/zhicp/data, the loop-closure use of CICPWarpper: MapNode.cpp:625-656)
    zd = os.path.join(ROOT, "tests", "golden", "zhicp")
    try:
        rows = np.loadtxt(os.path.join(zd, "matched_nodes.log"))
        ids = sorted({int(v) for row in rows for v in row[:2] if os.path.exists(os.path.join(zd, "%d.log" % int(v)))})
        cl = {i: np.loadtxt(os.path.join(zd, "%d.log" % i), dtype=np.float32) for i in ids}
        prs = [(ids.index(int(a)), ids.index(int(b))) for a, b, *_ in rows if int(a) in cl and int(b) in cl]
        if prs:
            xs = [cl[i][:, 0] * 100.0 for i in ids]; ys = [cl[i][:, 1] * 100.0 for i in ids]
            oks = [np.ones(len(v), bool) for v in xs]
            ctx = capi.Context(capi.param_preset("LMS211")); ctx.set_stream(torch.cuda.current_stream().cuda_stream)
            cnt = 1 << 12
            res["zhicp_node_clouds"] = p2p_entry(torch, capi, ctx, p2p_problems(xs, ys, oks, prs, cnt), cnt,
                                                 "CICPWarpper::ICPMatch on the reference's map-node clouds (src/zhicp/data), zero prior")
            ctx.close()
    except OSError:
        pass
    return res


def secondary_latency(torch, capi, raw1081):
    """Single-call latency of the drop-in matchers' C-ABI entry points - the call pattern of src/slam (one FMatch per scan
    under a mutex, threadLocal1.cpp:198-220): lsa_fmatch_host (CPolarMatch::FMatch: preprocessing of the new scan + pm_icp +
    pm_error_index2 through the pinned staging block) and lsa_plicp_host (CCanonicalMatcher::FMatch) over consecutive scans
    of the logs' heads and of the synthetic 1081-beam set.  Host buffers in, host results out: PCIe and the ctypes call
    (a few microseconds) are inside the figures."""
    from laser_slam_amd import synth
    res = {}
    cases = [("fr079", "LMS511", "fr079_head.log", "ROBOTLASER1"), ("intel", "LMS211", "intel_head.log", "ROBOTLASER1"),
             ("mit", "LMS511", "mit_head.log", "ROBOTLASER1"), ("indoor", "LMS151", "indoor_head.txt", "LMS151"), ("syn1081", None, None, None)]
    for name, preset, fname, tag in cases:
        if preset is None:
            p = capi.param_make(**synth.SYN1081); r = raw1081[:60]; bad = np.zeros(r.shape, np.int32)
        else:
            path = os.path.join(ROOT, "tests", "golden", fname)
            if not os.path.exists(path):
                continue
            p = capi.param_preset(preset)
            r, _, _, bad, _ = capi.carmen_read(p, path, tag)
        S, n = r.shape
        ctx = capi.Context(p)
        z = np.zeros(n, np.int32)
        _, _, _, cur = ctx.fmatch_host([r[0].copy(), bad[0].astype(np.int32), z.copy()], [r[1].copy(), bad[1].astype(np.int32), z.copy()], True)
        tf, tp = [], []
        for k in range(2, min(S, 60)):
            t0 = time.perf_counter()
            _, _, _, cur = ctx.fmatch_host(cur, [r[k].copy(), bad[k].astype(np.int32), z.copy()], False)
            tf.append(time.perf_counter() - t0)
        # a lone CPolarMatch::pm_psm (the public method, e.g. src/zhpsm/test.cpp) on preprocessed scans
        pre = [ctx.preprocess_host(r[k], bad[k].astype(np.int32)) for k in range(min(S, 30))]
        tq = []
        for k in range(1, len(pre)):
            t0 = time.perf_counter(); ctx.match_host(0, pre[k - 1], pre[k]); tq.append(time.perf_counter() - t0)
        capi.plicp_host(ctx, r[0], r[1])
        its = []
        for k in range(2, min(S, 60)):
            t0 = time.perf_counter()
            o = capi.plicp_host(ctx, r[k - 1], r[k])
            tp.append(time.perf_counter() - t0); its.append(o[5])
        ctx.close()
        q = lambda v: {"median_us": float(np.median(v)) * 1e6, "p95_us": float(np.percentile(v, 95)) * 1e6, "max_us": float(np.max(v)) * 1e6}
        res[name] = {"beams": n, "calls": len(tf), "CPolarMatch_FMatch": q(tf), "CPolarMatch_pm_psm": q(tq[3:]),
                     "CCanonicalMatcher_FMatch": dict(q(tp), mean_iterations=float(np.mean(its)), restarting_calls_frac=float(np.mean(np.array(its) > 10)))}
    q = lambda v: {"median_us": float(np.median(v)) * 1e6, "p95_us": float(np.percentile(v, 95)) * 1e6, "max_us": float(np.max(v)) * 1e6}
    # a lone CPMap::updateMap (drawmap / mapGraph insert a scan at a time): one 1081-beam scan into 4096^2 at 5 cm
    p = capi.param_make(**synth.SYN1081); ctx = capi.Context(p)
    fi = capi.param_tables(p)[0]
    g = capi.Grid(ctx, 4096, 4096, 2048, 2048, 0.0, 0.0, 0.05)
    tu = []
    for k in range(60):
        rr = raw1081[k] * 0.01
        x = (rr * np.cos(fi) + 1.0).astype(np.float32); y = (rr * np.sin(fi) + 2.0).astype(np.float32)
        t0 = time.perf_counter(); g.update_host(x, y, 1.0, 2.0); ctx.sync(); tu.append(time.perf_counter() - t0)
    g.close()
    res["CPMap_updateMap"] = dict(q(tu[10:]), beams=int(raw1081.shape[1]), grid="4096x4096 at 5 cm")
    # a lone CICPWarpper::ICPMatch on the reference's map-node clouds with their recorded priors (MapNode.cpp:625-656)
    zd = os.path.join(ROOT, "tests", "golden", "zhicp")
    try:
        rows = np.loadtxt(os.path.join(zd, "matched_nodes.log"))
        tc, itc, npts = [], [], []
        for rep in range(4):
            for a, b, _, px, py, pth in rows:
                fa, fb = os.path.join(zd, "%d.log" % int(a)), os.path.join(zd, "%d.log" % int(b))
                if not (os.path.exists(fa) and os.path.exists(fb)):
                    continue
                ca, cb = np.loadtxt(fa, dtype=np.float32), np.loadtxt(fb, dtype=np.float32)
                t0 = time.perf_counter()
                o = capi.p2p_icp_host(ctx, ca[:, 0].copy(), ca[:, 1].copy(), cb[:, 0].copy(), cb[:, 1].copy(), (px, py, pth))
                if rep:
                    tc.append(time.perf_counter() - t0); itc.append(o[5]); npts.append(len(ca) + len(cb))
        if tc:
            res["CICPWarpper_ICPMatch"] = dict(q(tc), mean_iterations=float(np.mean(itc)), points_per_problem=float(np.mean(npts)))
    except OSError:
        pass
    ctx.close()
    return {"fmatch_latency": res}


def secondary_grid(torch, capi, ctx, raw, p):
    """CPMap: binned batch insertion into 4096^2 @ 5 cm and the probability pass."""
    from laser_slam_amd import synth
    res = {}
    n = raw.shape[1]
    fi = capi.param_tables(p)[0]
    nb = 2048
    poses = synth.path_pose(np.arange(nb) % N_BASE)               # (scan s * N_BASE + k of the set was taken at path_pose(k))
    rr = np.concatenate([raw, raw])[:nb] / 100.0
    gx = (poses[:, 0:1] + rr * np.cos(poses[:, 2:3] + fi[None])).astype(np.float32)
    gy = (poses[:, 1:2] + rr * np.sin(poses[:, 2:3] + fi[None])).astype(np.float32)
    grid = capi.Grid(ctx, 4096, 4096, 1024, 1024, 0.0, 0.0, 0.05)
    dgx, dgy = torch.from_numpy(gx).cuda(), torch.from_numpy(gy).cuda()
    drob = torch.from_numpy(poses[:, :2].astype(np.float32)).cuda()
    ms = timed(torch, lambda: grid.update_batch(nb, n, dgx, dgy, drob), reps=2)
    cells = float(np.mean(np.maximum(np.abs(gx - poses[:, 0:1]), np.abs(gy - poses[:, 1:2])) / 0.05)) + 1
    res["grid_update"] = {"value": nb / (ms * 1e-3), "unit": "scans/s", "scans": nb, "cells_per_s": nb * n * cells / (ms * 1e-3),
                          "mean_ray_cells": cells,
                          "roofline": hbm_roofline("grid_binned_kernel", nb * n * (cells * 8 + 16), ms, nb,
                                                   "EFFECTIVE: SURVEY 8d's algorithmic read-modify-write bytes N (L 8 + 16) per scan over the "
                                                   "whole batch update (prep + binning + walk); the walk accumulates in LDS tiles, so the bytes "
                                                   "that reach HBM (`traffic`, PMC) are far fewer")}
    ms = timed(torch, lambda: grid.compute_prob(), reps=5)
    res["grid_prob_cached"] = {"value": 4096 * 4096 / (ms * 1e-3), "unit": "cells/s",
                               "roofline": {"bound": "infinity-cache", "achieved": 12.0 * 4096 * 4096 / (ms * 1e-3) / 1e9, "peak": None, "unit": "GB/s",
                                            "frac": None, "traffic": None, "kernel": "grid_prob_kernel", "kernel_ms": ms,
                                            "note": "12 B per cell; back-to-back launches: the 201 MB working set stays in the 256 MiB Infinity "
                                                    "Cache, so this is a cache figure and is not priced against HBM"}}
    # cold: a 1 GiB buffer is written between the launches, which evicts the layers from the Infinity Cache (and L2)
    flush = torch.empty(1 << 28, dtype=torch.float32, device="cuda")
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cold = []
    for _ in range(4):
        flush.fill_(1.0)
        a.record(); grid.compute_prob(); b.record(); torch.cuda.synchronize()
        cold.append(a.elapsed_time(b))
    ms = float(np.median(cold[1:]))
    del flush
    res["grid_prob"] = {"value": 4096 * 4096 / (ms * 1e-3), "unit": "cells/s",
                        "roofline": hbm_roofline("grid_prob_kernel", 12.0 * 4096 * 4096, ms, 1,
                                                 "12 B per cell (two u32 count layers read, the f32 probability layer written), layers evicted "
                                                 "from the Infinity Cache before every timed launch: the HBM figure")}
    grid.close()
    return res


def pf_world(torch, capi, ctx, raw, p):
    """Map, likelihood field and observed scan of the particle-filter entries: 256 synthetic scans inserted at their true
    poses into 4096^2 @ 5 cm, the field built from it, the first scan as laser-frame points."""
    from laser_slam_amd import synth
    n = raw.shape[1]
    fi = capi.param_tables(p)[0]
    nb = 256                                                  # (scan s * N_BASE + k of the set was taken at path_pose(k))
    poses = synth.path_pose(np.arange(nb) % N_BASE)
    rr = raw[np.arange(nb)] / 100.0
    gx = (poses[:, 0:1] + rr * np.cos(poses[:, 2:3] + fi[None])).astype(np.float32)
    gy = (poses[:, 1:2] + rr * np.sin(poses[:, 2:3] + fi[None])).astype(np.float32)
    G = 4096
    grid = capi.Grid(ctx, G, G, 1024, 1024, 0.0, 0.0, 0.05)
    grid.update_batch(nb, n, torch.from_numpy(gx).cuda(), torch.from_numpy(gy).cuda(), torch.from_numpy(poses[:, :2].astype(np.float32)).cuda())
    grid.compute_prob()
    fld = capi.Field(ctx, G, G, 1024, 1024, 0.0, 0.0, 0.05)
    fld.build_from_grid(grid); ctx.sync()
    sxy = (rr[0][:, None] * np.stack([np.cos(fi), np.sin(fi)], 1)).astype(np.float32)
    d_sx, d_sy = torch.from_numpy(np.ascontiguousarray(sxy[:, 0])).cuda(), torch.from_numpy(np.ascontiguousarray(sxy[:, 1])).cuda()
    return grid, fld, d_sx, d_sy, n


PF_PREDICT = ("predict", [10.0, 7.5, 0.0], [0.25, 0.25, 0.27])
PF_BYTES_PER_PARTICLE = 92.0     # predict 12 r + 12 w, score 12 r + 4 w, weights 4 r + 8 w, offspring 8 r + 12 r + 12 w, roughen 12 r + 12 r + 12 w (less the re-reads L2 serves)


def pf_cloud(torch, P, seed):
    rng = np.random.default_rng(seed)
    pp = np.stack([10 + rng.normal(0, 1.0, P), 7.5 + rng.normal(0, 1.0, P), rng.normal(0, 0.3, P)], 1).astype(np.float32)
    return torch.from_numpy(pp).cuda()


def secondary_global(torch, capi, ctx, raw, p):
    """Global localisation (BASELINE config 3: the brute-force (x, y, theta) pose search on synthetic 1081-beam scans; in the
    reference threadLocalization::globalization, src/localization/localization.cpp:483-540 - SURVEY.md 0 maps "zhcsm
    correlative" to it): candidate poses on randomPose's lattice over the 20 m x 15 m room -> CVPmap::isValidPose -> scorer
    -> topK(60).  Two sizes: 2^20 candidates with the likelihood-field scorer (obsLikelyhood3), and the reference's own
    10 000 candidates with its default scorer obsLikelyhood (a virtual scan ray-cast per pose, aligned with the observed one
    by CICPWarpper's ICP)."""
    from laser_slam_amd import localization, synth
    res = {}
    grid, fld, d_sx, d_sy, n = pf_world(torch, capi, ctx, raw, p)
    d_prob = torch.from_numpy(grid.download(2)).cuda()
    rng = np.random.default_rng(5)
    true = synth.path_pose(0).astype(np.float32)
    for name, P, scorer in (("likelihood_field_2^20", 1 << 20, "likelihood"), ("icp_goodness_10000", 10000, "icp")):
        poses = localization.random_poses(P, synth.ROOM_W, synth.ROOM_H, rng)
        poses[0] = true                                        # (the observed scan's own pose is among the candidates)
        d_poses = torch.from_numpy(poses).cuda()
        if scorer == "likelihood":                             # the synthetic scan itself, plain pose (+) point frame
            scan = dict(x=d_sx, y=d_sy, decimation=3, std_frame=True)
        else:                                                  # what the map shows from the true pose, in CVPmap's scan frame
            nb = 361
            r = torch.zeros(1, nb, device="cuda"); bad = torch.zeros(1, nb, dtype=torch.uint8, device="cuda")
            capi.simulate_scans(fld, d_prob, d_poses[:1].contiguous(), 1, nb, r, bad); ctx.sync()
            ang = np.float32(-3.141592654 / 4.0) + np.arange(nb, dtype=np.float32) * (np.float32(capi.SIM_FOV) / np.float32(nb - 1))
            rr = r[0].cpu().numpy()
            scan = dict(x=torch.from_numpy((rr * np.cos(ang)).astype(np.float32)).cuda(), y=torch.from_numpy((rr * np.sin(ang)).astype(np.float32)).cuda())
        torch.cuda.synchronize()
        best, idx, sc = localization.global_search(fld, d_prob, d_poses, scan, scorer=scorer)
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            best, idx, sc = localization.global_search(fld, d_prob, d_poses, scan, scorer=scorer)
        dt = (time.perf_counter() - t0) / reps
        res[name] = {"value": P / dt, "unit": "candidate poses/s", "candidates": P, "ms_per_search": dt * 1e3, "scorer": scorer,
                     "best_is_true_pose": bool(int(idx[0]) == 0), "best_pose_error_m": float(np.hypot(best[0] - true[0], best[1] - true[1]))}
    fld.close(); grid.close()
    return {"global_search": res}


def secondary_pf(torch, capi, ctx, dist, rank, world, raw, p, coll_dev):
    """Particle-filter localisation (config 4): the likelihood-field scorer alone, then whole steps - prediction, scores,
    best pose, weights / N_eff, fixed-point systematic resampling with the surplus exchange, roughening - in the fused
    form (four launches, one read-back; pf.py over torch.distributed = RCCL with several ranks):
      pf_step         weak scaling, 2^20 particles PER RANK;
      pf_step_strong  config 4 as written: 2^20 particles OVER the ranks (131 072 each at eight);
      pf_shard_131072 (one rank only) the shard one rank of eight owns, alone on one GPU - no collective in it."""
    from laser_slam_amd import pf
    res = {}
    t0 = time.perf_counter()
    grid, fld, d_sx, d_sy, n = pf_world(torch, capi, ctx, raw, p)
    res["field_build"] = {"value": timed(torch, lambda: fld.build_from_grid(grid), reps=3), "unit": "ms", "grid": "4096x4096"}
    nbm = (n + 2) // 3
    P = 1 << 20
    if rank == 0:
        d_p = pf_cloud(torch, P, 1); d_sc = torch.zeros(P, device="cuda")
        torch.cuda.synchronize()
        ms = timed(torch, lambda: fld.score_poses(d_sx, d_sy, n, 3, d_p, P, d_sc), reps=5)
        sp = load_profile("pmc_score_poses") or {}
        rq = sp.get("l2_requests_per_lane_beam")
        l2_gbs = rq * P * nbm * 128.0 / (ms * 1e-3) / 1e9 if rq else None
        res["score_poses"] = {"value": P / (ms * 1e-3), "unit": "poses/s", "poses": P, "beams_per_pose": nbm,
                              "roofline": {"bound": "l2", "achieved": l2_gbs, "peak": L2_PEAK_GBS, "unit": "GB/s",
                                           "frac": l2_gbs / L2_PEAK_GBS if l2_gbs else None, "traffic": None, "kernel": "pf_score_kernel<false, 16>",
                                           "kernel_ms": ms, "l2_requests_per_lane_beam": rq, "tcp_hit_rate": sp.get("tcp_hit_rate"),
                                           "tcc_hit_rate": sp.get("tcc_hit_rate"), "valu_busy_frac": sp.get("valu_busy_frac"),
                                           "wave_wait_frac": sp.get("wave_wait_frac"),
                                           "gathered_gbs": P * nbm * 4 / (ms * 1e-3) / 1e9,
                                           "note": "a gather from the 64 MiB likelihood table (4 x 8-cell tiles of 128 B), served by the XCDs' L2s "
                                                   "(TCC hit rate ~1: nothing of it is an HBM stream); achieved = L1 -> L2 requests per lane and beam "
                                                   "(rocprofv3 TCP_TCC_READ_REQ, profiles/%s_pmc_score_poses.json) x 128 B x lane-beams/s measured "
                                                   "here, against the 34.5 TB/s the guide measures for the L2s; gathered_gbs = the 4 useful bytes per "
                                                   "lane and beam" % PROFILE_ROUND}}

    def run(n_total, reps):
        """fused step over this process's ranks: (ms per step max over ranks, info, bytes rank 0 sent per step)"""
        first, cnt = pf.shard_bounds(n_total, rank, world)
        fp = pf.FusedDevicePrims(ctx, fld, cnt, d_sx, d_sy, n, 3)
        filt = pf.FusedParticleFilter(n_total, fp, resample_threshold=2.0, roughen_K=0.2)       # (threshold 2: every step resamples)
        d_cur = pf_cloud(torch, cnt, 1 + rank)
        g = torch.Generator(device="cuda"); g.manual_seed(7 + rank)
        d_nrm = torch.randn(cnt, 3, device="cuda", generator=g)
        info = None
        for it in range(reps + 2):
            if it == 2:
                torch.cuda.synchronize()
                if dist is not None:
                    dist.barrier()
                t1 = time.perf_counter()
            d_cur, info = filt.step(d_cur, 0.37, predict=PF_PREDICT, normals_predict=d_nrm, normals_roughen=d_nrm)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        dt = (time.perf_counter() - t1) / reps
        if dist is not None:
            t = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        sent = filt.bytes_sent / (reps + 2)
        fp.close()
        return dt, info, sent, cnt

    def entry(n_total, scaling, reps):
        dt, info, sent, cnt = run(n_total, reps)
        return {"value": n_total / dt, "unit": "particles/s", "ms_per_step": dt * 1e3, "particles_total": n_total,
                "particles_per_gpu": cnt, "n_gpus": world, "scaling": scaling, "resampled_every_step": bool(info["resampled"]),
                "bytes_sent_per_step_rank0": sent, "launches_per_step": 4, "host_readbacks_per_step": 1,
                "collectives_per_step": 0 if world == 1 else 4,
                "roofline": {"bound": "hbm", "achieved": n_total * PF_BYTES_PER_PARTICLE / dt / 1e9, "peak": HBM_PEAK_GBS * world, "unit": "GB/s",
                             "frac": n_total * PF_BYTES_PER_PARTICLE / dt / 1e9 / (HBM_PEAK_GBS * world), "traffic": None,
                             "kernel": "pf step (pf_score / pf_weights / pf_offspring / pf_roughen_keys)", "kernel_ms": dt * 1e3,
                             "note": "about 92 B of particle arrays per particle and step beside the field gather; the step is the "
                                     "likelihood-field scoring (score_poses above: a cache gather, not an HBM stream) plus three short "
                                     "launches and one read-back; with several ranks two all-gathers (8 B and 48 B per rank), the surplus "
                                     "exchange and one all-reduce of 24 B"}}

    res["pf_step"] = entry(P * world, "weak", 5)
    res["pf_step_strong"] = entry(P, "strong", 10)
    if world == 1:
        Ps = 1 << 17
        e = entry(Ps, "one shard of eight, alone on one GPU", 20)
        # the same step as the C ABI's one call (lsa_pf_filter_step: what a C++ host runs; no Python between the launches)
        flt = capi.PfFilter(fld, Ps)
        d_cur = pf_cloud(torch, Ps, 1); d_out = torch.zeros(Ps, 3, device="cuda"); d_nrm = torch.randn(Ps, 3, device="cuda")
        torch.cuda.synchronize()

        def one():
            flt.step(d_sx, d_sy, n, 3, d_cur, d_out, Ps, Ps, 0.37, resample_threshold=2.0, roughen_K=0.2, d_normals_roughen=d_nrm,
                     predict=PF_PREDICT, d_normals_predict=d_nrm)
        one(); torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(20):
            one()
        torch.cuda.synchronize()
        e["ms_per_step_c_abi"] = (time.perf_counter() - t1) / 20 * 1e3
        e["speedup_of_full_set_step"] = res["pf_step_strong"]["ms_per_step"] / e["ms_per_step"]
        e["note"] = ("ms_per_step = pf.py's phases over torch (what pf_step times); ms_per_step_c_abi = lsa_pf_filter_step; "
                     "speedup_of_full_set_step = one-GPU step over 2^20 particles / this: the compute side of strong scaling to "
                     "eight ranks, collectives excluded (no multi-GPU box here)")
        res["pf_shard_131072"] = e
        flt.close()
    fld.close(); grid.close()
    res["pf_entries_s"] = time.perf_counter() - t0
    return res


if __name__ == "__main__":
    sys.exit(main() or 0)
