"""Sharded particle-filter step: one process per GPU, the particles split into contiguous
blocks over the ranks, map and scan replicated.  This module is the exchange logic over torch.distributed (RCCL over
xGMI on GPUs, gloo in the CPU tests); the same steps are available to C++ hosts through the C ABI's `lsa_comm`
entry points (include/lsa_core.h, csrc/shard_host.inc), which use RCCL directly.

Exchange steps of one step (messages are tiny except the last one):
  * best pose   all-gather of one (score, global index) record per rank -> exact arg-max (ties: smaller index)
  * weights     all-reduce(SUM) of {sum w, sum w^2}, w = exp(score - global max) -> N_eff
  * resampling  (only when N_eff is low) systematic resampling (src/sensorFusion/SIRFlt.cpp:118-191) in fixed point:
      - all-gather of ONE 8-byte weight sum per rank -> every rank knows the total, the stratum offset U0 and
        which offspring [E_r, E_{r+1}) each shard's particles are the parents of (`capi.pf_plan`, host arithmetic);
      - every rank resolves its own strata from its LOCAL cumulative sum (no weights, no cumulative sums and no
        particle array cross ranks);
      - one all-to-all moves the offspring that belong to another rank's block of the new particle array
        (12 B each).  Smooth weights leave almost everything where it is; bytes sent per rank <= 12 x the number of
        offspring the rank produced.
    Fixed point (q = floor(w 2^32), exact 64-bit sums) makes every number of the scheme independent of how the
    particles are split, so the sharded step reproduces the single-process step exactly (csrc/shard_kernels.hip).

Compute = HIP kernels behind the C ABI, handed in as `prims` (see `DevicePrims`); the CPU tests pass numpy stand-ins.
"""
import numpy as np
import torch
import torch.distributed as dist


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n_total, rank, world):
    """Contiguous block [first, first+count) of rank; blocks differ by at most one (== lsa_shard_bounds)."""
    base, rem = divmod(n_total, world)
    first = rank * base + min(rank, rem)
    return first, base + (1 if rank < rem else 0)


def global_best(local_max, local_index, first, device="cpu"):
    """Exact global arg-max from per-rank (max score, local index): the larger score wins,
    ties go to the smaller global index (the reference's topK scans from the front,
    src/localization/localization.cpp:126-149)."""
    rank, world = _world()
    rec = torch.tensor([float(local_max), float(first + local_index)], dtype=torch.float64, device=device)
    if world == 1:
        return float(rec[0]), int(rec[1])
    out = [torch.zeros_like(rec) for _ in range(world)]
    dist.all_gather(out, rec)
    out = torch.stack(out).cpu()
    best = max(range(world), key=lambda r: (float(out[r, 0]), -float(out[r, 1])))
    return float(out[best, 0]), int(out[best, 1])


def global_weight_sums(local_sum, local_sumsq, device="cpu"):
    """all-reduce(SUM) of {sum w, sum w^2}; returns (sum, sumsq, n_eff).  n_eff follows
    CParticles::neff (src/localization/particles.cpp:155-171, sic: the accumulator starts at 1)."""
    rank, world = _world()
    t = torch.tensor([local_sum, local_sumsq], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    s, s2 = float(t[0]), float(t[1])
    return s, s2, 1.0 / (1.0 + s2 / (s * s))


def gather_concat(t):
    """all-gather of equally- or unequally-sized shards along dim 0 (pads to the largest).  Used by tests and result
    collection only - the filter step itself never gathers particle arrays."""
    rank, world = _world()
    if world == 1:
        return t
    n = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    ns = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(ns, n)
    ns = [int(v) for v in ns]
    big = max(ns)
    pad = torch.zeros((big,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    outs = [torch.zeros_like(pad) for _ in range(world)]
    dist.all_gather(outs, pad)
    return torch.cat([o[:k] for o, k in zip(outs, ns)], 0)


def exchange_plan(bounds, n_total, rank, world):
    """From the offspring ranges [bounds[r], bounds[r+1]) the shards produce and the destination blocks of the new
    particle array: how many offspring this rank sends to / receives from every rank (in rank order, which is also
    ascending offspring order on both sides)."""
    my_first, my_count = shard_bounds(n_total, rank, world)
    p_lo, p_hi = int(bounds[rank]), int(bounds[rank + 1])
    send, recv = [], []
    for q in range(world):
        qf, qc = shard_bounds(n_total, q, world)
        send.append(max(0, min(p_hi, qf + qc) - max(p_lo, qf)))
        recv.append(max(0, min(int(bounds[q + 1]), my_first + my_count) - max(int(bounds[q]), my_first)))
    return send, recv


class DevicePrims:
    """The per-rank compute of the step on the GPU: thin calls into the C ABI (laser_slam_amd.capi)."""

    def __init__(self, ctx, field, d_scan_x, d_scan_y, n_points, decimation=3):
        from . import capi
        self.capi, self.ctx, self.field = capi, ctx, field
        self.sx, self.sy, self.n_points, self.decim = d_scan_x, d_scan_y, n_points, decimation

    def score(self, poses):
        s = torch.empty(poses.shape[0], device=poses.device)
        self.field.score_poses(self.sx, self.sy, self.n_points, self.decim, poses, poses.shape[0], s)
        return s

    def argmax(self, s):
        return self.capi.pose_argmax(self.ctx, s, s.shape[0])

    def weights(self, s, gmax):
        w = torch.empty_like(s)
        a, b = self.capi.pf_weights(self.ctx, s, s.shape[0], gmax, w)
        return w, a, b

    def fixed_scan(self, w):
        cum = torch.empty(w.shape[0], dtype=torch.int64, device=w.device)
        return cum, self.capi.pf_fixed_scan(self.ctx, w, w.shape[0], cum)

    def offspring_bounds(self, cum, offset, n_total, U0, W):
        e = torch.empty(cum.shape[0], dtype=torch.int32, device=cum.device)
        self.capi.pf_offspring_bounds(self.ctx, cum, cum.shape[0], offset, n_total, U0, W, e)
        return e

    def parents(self, e, m_first, m_count):
        p = torch.empty(m_count, dtype=torch.int32, device=e.device)
        self.capi.pf_parents(self.ctx, e, e.shape[0], m_first, m_count, p)
        return p

    def gather(self, poses, parent):
        out = torch.empty(parent.shape[0], 3, device=poses.device)
        self.capi.gather_poses(self.ctx, poses, parent, parent.shape[0], out)
        self.ctx.sync()
        return out


class ShardedParticleFilter:
    """One localisation step over this rank's particle shard; `prims` supplies score / argmax / weights /
    fixed_scan / offspring_bounds / parents / gather (DevicePrims, or stand-ins in the CPU tests)."""

    def __init__(self, n_total, prims, resample_threshold=0.5, device="cpu"):
        self.rank, self.world = _world()
        self.n_total = n_total
        self.first, self.count = shard_bounds(n_total, self.rank, self.world)
        self.prims = prims
        self.resample_threshold = resample_threshold          # g_resample_threshold, globaldef.cpp:22
        self.device = device
        self.bytes_sent = 0                                   # payload handed to other ranks (all steps so far)

    def resample(self, w, poses, u0):
        """Systematic resampling of the sharded set; returns this rank's block of the new particle array."""
        from . import capi
        P, rank, world = self.prims, self.rank, self.world
        cum, total = P.fixed_scan(w)                         # exact cumulative fixed-point weights of the shard
        mine = torch.tensor([total], dtype=torch.int64, device=self.device)               # < 2^63
        if world > 1:
            outs = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(outs, mine)
            sums = [int(v) for v in outs]
            self.bytes_sent += 8 * (world - 1)
        else:
            sums = [int(mine)]
        W, U0, bounds = capi.pf_plan(sums, self.n_total, u0)
        offset = sum(sums[:rank])
        p_lo, p_cnt = int(bounds[rank]), int(bounds[rank + 1] - bounds[rank])
        if p_cnt > 0:
            e = P.offspring_bounds(cum, offset, self.n_total, U0, W)
            prod = P.gather(poses, P.parents(e, p_lo, p_cnt))                                # offspring p_lo .. p_lo + p_cnt - 1
        else:
            prod = poses[:0]
        if world == 1:
            return prod.contiguous()
        send, recv = exchange_plan(bounds, self.n_total, rank, world)
        out = torch.empty(self.count, 3, dtype=poses.dtype, device=poses.device)
        if dist.get_backend() == "gloo" and prod.is_cuda:    # (rehearsals with ranks sharing a GPU: gloo moves host memory)
            out_h = torch.empty(self.count, 3, dtype=poses.dtype)
            dist.all_to_all_single(out_h, prod.cpu().contiguous(), output_split_sizes=recv, input_split_sizes=send)
            out.copy_(out_h)
        else:
            dist.all_to_all_single(out, prod.contiguous(), output_split_sizes=recv, input_split_sizes=send)
        self.bytes_sent += 12 * (sum(send) - send[rank])
        return out

    def step(self, poses, u0):
        """poses: this rank's [count,3] tensor.  Returns (new poses, info dict)."""
        P = self.prims
        s = P.score(poses)
        lmax, lidx = P.argmax(s)
        gmax, gidx = global_best(lmax, lidx, self.first, self.device)
        w, lsum, lsumsq = P.weights(s, gmax)
        tot, tot2, neff = global_weight_sums(lsum, lsumsq, self.device)
        if self.world > 1:
            self.bytes_sent += (16 + 16) * (self.world - 1)
        info = dict(best_score=gmax, best_index=gidx, weight_sum=tot, n_eff=neff, resampled=False)
        if neff < self.resample_threshold * self.n_total:      # particles.cpp:350
            poses = self.resample(w, poses, u0)
            info["resampled"] = True
        return poses, info
