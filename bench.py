#!/usr/bin/env python3
"""Headline benchmark: scan-matches/s on synthetic 1081-beam scans (BASELINE.json).

A "step" = one batch of `--pairs` scan-pair matches (pm_icp, the matcher behind the
reference's FMatch front door) over a device-resident, already preprocessed scan set.
Inputs are in HBM before the timed region.  With N ranks (torch.distributed / RCCL)
each rank matches its own batch (independent pairs -> no data-path collective, weak
scaling); value = all pairs of all ranks / max-over-ranks time.

Prints ONE JSON line (rank 0) with the contract's fields plus `roofline`
(dominant kernel = icp_kernel, algorithmic bytes per SURVEY.md §8d) and
`cpu_baseline` (the reference built in place when oracle/_ref travelled, else the
oracle port; bounded sample on the host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BYTES_PER_PAIR = 2 * 1081 * 4 + 12 + 20      # 8 680 B per 1081-beam pair
HBM_PEAK_GBS = 8000.0                        # MI355X_MICROARCH.md: 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3


def build_scan_set(n_scans, seeds):
    from laser_slam_amd import synth
    sets = [synth.make_scans(n_scans, noise_seed=synth.WORLD_SEED + j)[0] for j in range(seeds)]
    return np.concatenate(sets, 0)


def cpu_baseline(raw, n_pairs, which, gpu_match=None):
    """Time the CPU path on a bounded sample of the same workload (rank 0 only).  With `gpu_match` the device's
    poses for the very same pairs are set against the CPU's: BASELINE.json's "pose RMSE vs CPU ref"."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    from laser_slam_amd import synth
    use_ref = oracle_lib.have_ref()
    eng = oracle_lib.Ref(**synth.SYN1081) if use_ref else oracle_lib.Oracle(**synth.SYN1081)
    pre = [eng.preprocess(raw[k]) for k in range(n_pairs + 1)]
    t0 = time.perf_counter()
    cpu_pose = np.zeros((n_pairs, 3), np.float32); cpu_st = np.zeros(n_pairs, np.int32)
    for b in range(n_pairs):
        res = eng.match(which, pre[b], pre[b + 1])
        cpu_st[b] = res[0]; cpu_pose[b] = res[1]
    dt = time.perf_counter() - t0
    out = {"value": n_pairs / dt, "unit": "scan-matches/s", "cores": 1,
           "kind": "reference" if use_ref else "port",
           "sample": "%d consecutive synthetic 1081-beam pairs, pm_icp, single thread, preprocessing excluded (%.1f s)"
                     % (n_pairs, dt)}
    if gpu_match is not None:
        g_pose, g_st = gpu_match(n_pairs)
        ok = (cpu_st == 0) & (g_st == 0)
        d = (g_pose[ok].astype(np.float64) - cpu_pose[ok].astype(np.float64))
        out["pose_rmse_vs_cpu"] = {
            "pairs": int(n_pairs), "matched_by_both": int(ok.sum()), "status_mismatches": int(((cpu_st == 0) != (g_st == 0)).sum()),
            "xy_m": float(np.sqrt(np.mean(d[:, 0] ** 2 + d[:, 1] ** 2)) * 0.01) if ok.any() else None,
            "theta_rad": float(np.sqrt(np.mean(d[:, 2] ** 2))) if ok.any() else None,
            "bit_identical": bool(np.array_equal(g_pose[ok].view(np.uint32), cpu_pose[ok].view(np.uint32)))}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=1 << 20, help="scan pairs per step per GPU (SURVEY 8(d) batch sizes: 1 .. 2^20)")
    ap.add_argument("--matcher", default="icp", choices=["icp", "psm"])
    ap.add_argument("--cpu-pairs", type=int, default=400)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--extras", action="store_true", help="also time PSM, preprocess and grid update")
    args = ap.parse_args()

    import torch
    from laser_slam_amd import capi, synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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

    # ---- workload: synthetic 1081-beam scans, consecutive pairs, zero initial guess ----
    n_base, seeds = 200, 8
    raw = build_scan_set(n_base, seeds)                       # [1600][1081]
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
    base_pairs = np.array([(s * n_base + k - 1, s * n_base + k) for s in range(seeds) for k in range(1, n_base)], np.int32)
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
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    mean_iters = float(it.float().mean().item())
    fail_frac = float((st != 0).float().mean().item())
    total_pairs = B * args.steps * world
    value = total_pairs / elapsed

    out = None
    if rank == 0:
        achieved = BYTES_PER_PAIR * B / (kernel_ms * 1e-3) / 1e9
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
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "%s_kernel" % args.matcher, "kernel_ms": kernel_ms,
                         "bytes_per_pair": BYTES_PER_PAIR,
                         "note": "the matcher is vector-instruction-issue bound, not HBM bound (DESIGN.md 4.1; valu_busy_frac); "
                                 "effective_fp32_frac prices the REFERENCE algorithm's arithmetic (brute-force window "
                                 "search) at this rate against the fp32 vector peak - the kernel prunes that search "
                                 "exactly, so these are not executed flops",
                         "effective_fp32_frac": flops_pair * B / (kernel_ms * 1e-3) / (FP32_PEAK_TFLOPS * 1e12),
                         "reference_flops_per_pair": flops_pair},
        }
        out["secondary"] = secondary(torch, capi, ctx, d_r, d_bad, d_seg, d_ri, d_ai, B, which)
        out["roofline"]["traffic"] = pmc_traffic(args.matcher, B)
        out["roofline"]["valu_busy_frac"] = pmc_valu_busy(args.matcher)
        if args.extras:
            out["extras"] = extras(torch, capi, ctx, d_r, d_bad, d_seg, d_ri, d_ai, B, raw, p)
        if not args.no_cpu and world == 1:                   # (rank 0 at N = 1 only: the other ranks would wait for it)
            def gpu_match(m):                              # the CPU sample's pairs (k, k + 1) on the device
                ri = torch.arange(0, m, dtype=torch.int32, device="cuda"); ai = ri + 1
                gp = torch.zeros(m, 3, device="cuda"); gs = torch.zeros(m, dtype=torch.int32, device="cuda")
                torch.cuda.synchronize()
                ctx.match_batch(which, m, d_r, d_bad, d_seg, ri, ai, gp, d_status=gs); ctx.sync()
                return gp.cpu().numpy(), gs.cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(raw, args.cpu_pairs, 1 if which == capi.MATCH_ICP else 0, gpu_match)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def secondary(torch, capi, ctx, d_r, d_bad, d_seg, d_ri, d_ai, B, which):
    """The other matcher of the path on the same batch (one timed launch), for the record."""
    other = capi.MATCH_PSM if which == capi.MATCH_ICP else capi.MATCH_ICP
    pose = torch.zeros(B, 3, device="cuda")
    ctx.match_batch(other, B, d_r, d_bad, d_seg, d_ri, d_ai, pose)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); ctx.match_batch(other, B, d_r, d_bad, d_seg, d_ri, d_ai, pose); b.record()
    torch.cuda.synchronize()
    return {("pm_psm" if other == capi.MATCH_PSM else "pm_icp") + "_matches_per_s": B / (a.elapsed_time(b) * 1e-3)}


def pmc_traffic(matcher, B):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/r01_pmc_hbm_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate --pmc runs at
    32768 pairs per launch; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), scaled to
    this batch size.  None when the profile file is absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")
    try:
        rec = json.load(open(path))["lsa::%s_kernel" % matcher]
        per_pair = (2.0 * rec["FETCH_SIZE_KB_per_launch"] + rec["WRITE_SIZE_KB_per_launch"]) * 1024.0 / rec.get("pairs_per_launch", 32768)
        return per_pair * B
    except Exception:
        return None


def pmc_valu_busy(matcher):
    """Share of the SIMDs' issue slots the dominant kernel keeps busy with vector instructions, from the
    committed SQ counter pass (profiles/r01_pmc_sq_counters.json): SQ_ACTIVE_INST_VALU (quad-cycles) over
    GRBM_GUI_ACTIVE / 8 XCDs / 4 cycles x 1024 SIMDs.  This, not HBM, is the resource the matcher runs
    against.  None when the profile file is absent."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_sq_counters.json")
    try:
        rec = json.load(open(path))["lsa::%s_kernel" % matcher]
        return rec["SQ_ACTIVE_INST_VALU"] / (rec["GRBM_GUI_ACTIVE"] / 8.0 / 4.0 * 1024.0)
    except Exception:
        return None


def extras(torch, capi, ctx, d_r, d_bad, d_seg, d_ri, d_ai, B, raw, p):
    """Secondary kernels of the path, timed the same way (not the headline)."""
    from laser_slam_amd import synth
    res = {}
    # point-to-line ICP (CCanonicalMatcher / zhcsm's matcher), raw scans, 8192 pairs
    Bp = 8192
    d_raw = torch.from_numpy(raw).cuda()
    d_o8 = torch.zeros(Bp, 8, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    capi.plicp_batch(ctx, Bp, d_raw, d_ri[:Bp].contiguous(), d_ai[:Bp].contiguous(), d_o8); ctx.sync()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); capi.plicp_batch(ctx, Bp, d_raw, d_ri[:Bp].contiguous(), d_ai[:Bp].contiguous(), d_o8); b.record()
    torch.cuda.synchronize()
    res["plicp_matches_per_s"] = Bp / (a.elapsed_time(b) * 1e-3)
    res["plicp_mean_iterations"] = float(d_o8[:, 5].mean().item())
    pose = torch.zeros(B, 3, device="cuda")

    def timed(fn, reps=3):
        fn(); torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps

    ms = timed(lambda: ctx.match_batch(capi.MATCH_PSM, B, d_r, d_bad, d_seg, d_ri, d_ai, pose))
    res["pm_psm_matches_per_s"] = B / (ms * 1e-3)
    S, n = raw.shape
    Sp = 16384                                              # batched preprocessing (one lane per scan from 4096 scans on)
    big = np.tile(raw, ((Sp + S - 1) // S, 1))[:Sp]
    r2 = torch.from_numpy(big).cuda(); b2 = torch.zeros(Sp, n, dtype=torch.uint8, device="cuda"); s2 = torch.zeros(Sp, n, dtype=torch.int16, device="cuda")
    ms = timed(lambda: ctx.preprocess_batch(Sp, r2, b2, s2), reps=1)
    res["preprocess_scans_per_s"] = Sp / (ms * 1e-3)
    # grid update: 4096^2 @ 5 cm, scans inserted at their true poses
    fi = capi.param_tables(p)[0]
    nb = 2048
    _, poses = synth.make_scans(1)
    poses = synth.path_pose(np.arange(nb))
    rr = np.concatenate([raw, raw])[:nb] / 100.0
    gx = (poses[:, 0:1] + rr * np.cos(poses[:, 2:3] + fi[None])).astype(np.float32)
    gy = (poses[:, 1:2] + rr * np.sin(poses[:, 2:3] + fi[None])).astype(np.float32)
    grid = capi.Grid(ctx, 4096, 4096, 1024, 1024, 0.0, 0.0, 0.05)
    dgx, dgy = torch.from_numpy(gx).cuda(), torch.from_numpy(gy).cuda()
    drob = torch.from_numpy(poses[:, :2].astype(np.float32)).cuda()
    ms = timed(lambda: grid.update_batch(nb, n, dgx, dgy, drob), reps=2)
    cells = float(np.mean(np.maximum(np.abs(gx - poses[:, 0:1]), np.abs(gy - poses[:, 1:2])) / 0.05)) + 1
    res["grid_update_scans_per_s"] = nb / (ms * 1e-3)
    res["grid_update_cells_per_s"] = nb * n * cells / (ms * 1e-3)
    res["grid_update_GBps_algorithmic"] = nb * n * (cells * 8 + 16) / (ms * 1e-3) / 1e9
    ms = timed(lambda: grid.compute_prob(), reps=3)
    res["grid_prob_GBps"] = 12 * 4096 * 4096 / (ms * 1e-3) / 1e9
    # pose scoring (likelihood field of the map just built): 1 M poses around the path, 1081-beam scan / 3
    fld = capi.Field(ctx, 4096, 4096, 1024, 1024, 0.0, 0.0, 0.05)
    t0 = time.perf_counter()
    fld.build_from_grid(grid); ctx.sync()
    res["field_build_ms_4096sq"] = (time.perf_counter() - t0) * 1e3
    P = 1 << 20
    rng = np.random.default_rng(1)
    pp = np.stack([10 + rng.normal(0, 1.0, P), 7.5 + rng.normal(0, 1.0, P), rng.normal(0, 0.3, P)], 1).astype(np.float32)
    d_p = torch.from_numpy(pp).cuda()
    sxy = (rr[0][:, None] * np.stack([np.cos(fi), np.sin(fi)], 1)).astype(np.float32)
    d_sx, d_sy = torch.from_numpy(np.ascontiguousarray(sxy[:, 0])).cuda(), torch.from_numpy(np.ascontiguousarray(sxy[:, 1])).cuda()
    d_sc = torch.zeros(P, device="cuda")
    torch.cuda.synchronize()
    ms = timed(lambda: fld.score_poses(d_sx, d_sy, n, 3, d_p, P, d_sc), reps=3)
    res["score_poses_per_s"] = P / (ms * 1e-3)
    res["score_gather_GBps"] = P * ((n + 2) // 3) * 4 / (ms * 1e-3) / 1e9
    # beam model (ray-cast 541-beam virtual scans in the map just built, 16384 poses) and top-60 of 1 M scores
    P2, nb2 = 16384, 541
    d_prob = torch.from_numpy(grid.download(2)).cuda()
    d_or = torch.full((nb2,), 4.0, device="cuda"); d_ob = torch.zeros(nb2, dtype=torch.uint8, device="cuda")
    d_s2 = torch.zeros(P2, device="cuda")
    torch.cuda.synchronize()
    ms = timed(lambda: capi.beam_model_scores(fld, d_prob, d_p, P2, d_or, d_ob, nb2, d_s2), reps=2)
    res["beam_model_poses_per_s"] = P2 / (ms * 1e-3)
    res["beam_model_rays_per_s"] = P2 * nb2 / (ms * 1e-3)
    # ICP-goodness scorer (obsLikelyhood): virtual 541-beam scan + point-to-point ICP per pose, 16384 poses
    P3 = 16384
    ang = (-3.141592654 / 4.0 + np.arange(nb2) * (4.712388981 / (nb2 - 1))).astype(np.float32)
    d_r1 = torch.zeros(1, nb2, device="cuda"); d_b1 = torch.zeros(1, nb2, dtype=torch.uint8, device="cuda")
    d_c = torch.tensor([[10.0, 7.5, 0.0]], device="cuda")
    torch.cuda.synchronize()
    capi.simulate_scans(fld, d_prob, d_c, 1, nb2, d_r1, d_b1); ctx.sync()
    r1 = d_r1.cpu().numpy()[0]                      # the scan seen from the centre of the pose cloud
    d_ox = torch.from_numpy((r1 * np.cos(ang)).astype(np.float32)).cuda(); d_oy = torch.from_numpy((r1 * np.sin(ang)).astype(np.float32)).cuda()
    d_w3 = torch.zeros(P3, device="cuda")
    torch.cuda.synchronize()
    ms = timed(lambda: capi.icp_goodness_scores(fld, d_prob, d_p, P3, d_ox, d_oy, nb2, d_w3), reps=1)
    res["icp_goodness_poses_per_s"] = P3 / (ms * 1e-3)
    t0 = time.perf_counter()
    capi.pose_topk(ctx, d_sc, P, 60)
    res["topk60_of_1M_ms"] = (time.perf_counter() - t0) * 1e3
    # the particle filter's resampling step at 1 M particles (weights, the reference's standard resampler, gather)
    m, _ = capi.pose_argmax(ctx, d_sc, P)
    d_w = torch.zeros(P, device="cuda"); d_par = torch.zeros(P, dtype=torch.int32, device="cuda"); d_new = torch.zeros(P, 3, device="cuda")
    d_u = torch.rand(P, dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()

    def resample_step():
        capi.pf_weights(ctx, d_sc, P, m, d_w)
        capi.pf_resample_standard(ctx, d_w, P, d_u, P, d_par)
        capi.gather_poses(ctx, d_p, d_par, P, d_new)
    resample_step(); ctx.sync()
    t0 = time.perf_counter(); resample_step(); ctx.sync()
    res["pf_resample_1M_ms"] = (time.perf_counter() - t0) * 1e3
    # one whole localisation step at 1 M particles as CParticles runs it (particles.cpp:119-153, 321-387): predict,
    # likelihood-field scores, best pose, weights / N_eff, standard resampling, gather, roughening, estimate of the 8 best
    d_nrm = torch.randn(P, 3, device="cuda"); d_cur = d_p.clone()

    def pf_step():
        capi.pf_predict(ctx, P, [10.0, 7.5, 0.0], [0.25, 0.25, 0.27], d_nrm, d_cur)
        fld.score_poses(d_sx, d_sy, n, 3, d_cur, P, d_sc)
        mm, _ = capi.pose_argmax(ctx, d_sc, P)
        s1, s2 = capi.pf_weights(ctx, d_sc, P, mm, d_w)
        if 1.0 / (1.0 + s2 / (s1 * s1)) < 0.5 * P:
            capi.pf_resample_standard(ctx, d_w, P, d_u, P, d_par)
            capi.gather_poses(ctx, d_cur, d_par, P, d_new)
            capi.pf_roughen(ctx, P, 0.2, d_nrm, d_new)
        return capi.pf_estimate(ctx, d_cur, d_sc, P, 8)
    pf_step(); ctx.sync()
    t0 = time.perf_counter(); pf_step(); ctx.sync()
    res["pf_step_1M_ms"] = (time.perf_counter() - t0) * 1e3
    fld.close()
    grid.close()
    return res


if __name__ == "__main__":
    main()
