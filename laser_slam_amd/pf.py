"""Sharded pose scoring / particle-filter step: one process per GPU,
particles split evenly over the ranks, map + scan replicated.

Exchange steps (torch.distributed; RCCL over xGMI on GPUs, gloo in the CPU tests):
  * best pose: all-gather of one (score, global index) record per rank -> exact arg-max
  * weights:   all-reduce(SUM) of {sum w, sum w^2} (w = exp(score - global max)) -> N_eff
  * resample:  all-gather of the weights and poses; every rank draws its own contiguous
    block of the systematic grid (src/sensorFusion/SIRFlt.cpp:118-191) from the same
    cumulative weights, so the result equals single-process systematic resampling.  The reference's
    default Standard_resampler (SIRFlt.cpp:39-115, one uniform per offspring) shards the same way:
    pass the shared uniforms as `u0` and a `local_resample` built on lsa_pf_resample_standard, whose
    (first, M) block of the parent-ascending list is this rank's share.

The collectives carry a few bytes (records) or 16 B per particle (resampling); the
scoring itself never leaves the GPU.  Compute = HIP kernels behind the C ABI
(`local_*` callables); this module is the host-side exchange logic only, so it can be
exercised on CPU ranks with stand-in local results.
"""
import numpy as np
import torch
import torch.distributed as dist


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n_total, rank, world):
    """Contiguous block [first, first+count) of rank; blocks differ by at most one."""
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
    best = max(out, key=lambda r: (float(r[0]), -float(r[1])))
    return float(best[0]), int(best[1])


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
    """all-gather of equally- or unequally-sized shards along dim 0 (pads to the largest)."""
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


class ShardedParticleFilter:
    """One localisation step over this rank's particle shard.

    local_score(poses[M,3]) -> scores[M]      (lsa_score_poses)
    local_argmax(scores) -> (max, index)      (lsa_pose_argmax)
    local_weights(scores, gmax) -> (w[M], sum, sumsq)     (lsa_pf_weights)
    local_resample(all_w[N], u0, first, M) -> parent[M]   (lsa_pf_resample)
    local_gather(all_poses[N,3], parent[M]) -> poses[M,3] (lsa_gather_poses)
    """

    def __init__(self, n_total, local_score, local_argmax, local_weights, local_resample, local_gather,
                 resample_threshold=0.5, device="cpu"):
        self.rank, self.world = _world()
        self.n_total = n_total
        self.first, self.count = shard_bounds(n_total, self.rank, self.world)
        self.fn = (local_score, local_argmax, local_weights, local_resample, local_gather)
        self.resample_threshold = resample_threshold          # g_resample_threshold, globaldef.cpp:22
        self.device = device

    def step(self, poses, u0):
        """poses: this rank's [count,3] tensor.  Returns (new poses, info dict)."""
        score, argmax, weights, resample, gather = self.fn
        s = score(poses)
        lmax, lidx = argmax(s)
        gmax, gidx = global_best(lmax, lidx, self.first, self.device)
        w, lsum, lsumsq = weights(s, gmax)
        tot, tot2, neff = global_weight_sums(lsum, lsumsq, self.device)
        info = dict(best_score=gmax, best_index=gidx, weight_sum=tot, n_eff=neff, resampled=False)
        if neff < self.resample_threshold * self.n_total:      # particles.cpp:350
            all_w = gather_concat(w)
            all_p = gather_concat(poses)
            parent = resample(all_w, u0, self.first, self.count)
            poses = gather(all_p, parent)
            info["resampled"] = True
        return poses, info
