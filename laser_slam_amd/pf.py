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

`FusedParticleFilter` is the same step cut to four launches, one read-back and - with several ranks - two small
all-gathers (the 8-byte best keys, the 48-byte records), the surplus exchange and one all-reduce of six extremes
(csrc/pfstep_kernels.hip, `lsa_pf_filter_*`); `ShardedParticleFilter` keeps the call-by-call form.
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


# ---- the fused step ---------------------------------------------------------------------------------------------
RECORD = np.dtype([("q_sum", "<u8"), ("w_sum", "<f8"), ("w2_sum", "<f8"), ("best_key", "<u8"), ("best_pose", "<f4", (3,)),
                   ("has_best", "<u4")])          # lsa::PfRecord, LSA_PF_RECORD_BYTES = 48
assert RECORD.itemsize == 48


def key_of(score, global_index):
    """ordered(score) << 32 | ~index: the larger key is the larger score, equal scores the smaller index"""
    u = int(np.float32(score).view(np.uint32))
    o = (~u & 0xFFFFFFFF) if u & 0x80000000 else (u | 0x80000000)
    return (o << 32) | (0xFFFFFFFF - int(global_index))


def key_score_index(key):
    o = (int(key) >> 32) & 0xFFFFFFFF
    bits = (o & 0x7FFFFFFF) if o & 0x80000000 else (~o & 0xFFFFFFFF)
    return float(np.uint32(bits).view(np.float32)), 0xFFFFFFFF - (int(key) & 0xFFFFFFFF)


def parse_records(raw, n_total):
    """raw: the ranks' records as bytes / int64 array (rank order) -> (info dict, fixed-point weight sum per rank).
    The same arithmetic as lsa_pf_filter_read: sums added in rank order, N_eff as CParticles::neff (sic)."""
    recs = np.frombuffer(np.ascontiguousarray(raw).tobytes(), dtype=RECORD)
    s = s2 = 0.0
    pose = None
    for r in recs:
        s += float(r["w_sum"]); s2 += float(r["w2_sum"])
        if r["has_best"]:
            pose = tuple(float(v) for v in r["best_pose"])
    score, index = key_score_index(int(recs[0]["best_key"]))
    info = dict(best_score=score, best_index=index, best_pose=pose, weight_sum=s, weight_sq_sum=s2,
                n_eff=1.0 / (1.0 + s2 / (s * s)), resampled=False)
    return info, [int(r["q_sum"]) for r in recs]


def f_ord(v):
    """order-preserving float32 -> uint32 map (csrc/lsa_device.h f_ord), vectorised"""
    u = np.ascontiguousarray(v, np.float32).view(np.uint32)
    return np.where(u & 0x80000000, ~u, u | np.uint32(0x80000000)).astype(np.uint32)


def ord_f(o):
    o = np.asarray(o, np.uint32)
    return np.where(o & 0x80000000, o & np.uint32(0x7FFFFFFF), ~o).astype(np.uint32).view(np.float32)


def minmax_keys(poses):
    """six int32 keys of a pose block as pf_offspring_kernel leaves them: [0:3] minima (complemented), [3:6] maxima, both
    sign-flipped so that a signed MAX combines blocks; an empty block gives the identity"""
    out = np.full(6, -2 ** 31, np.int64)
    p = np.asarray(poses, np.float32).reshape(-1, 3)
    if p.shape[0]:
        k = f_ord(p)
        out[3:] = (k.max(0) ^ np.uint32(0x80000000)).view(np.int32)
        out[:3] = ((~k).max(0) ^ np.uint32(0x80000000)).view(np.int32)
    return out.astype(np.int32)


def decode_minmax(mm):
    """-> (min[3], max[3]) float32 of the keys above"""
    k = np.asarray(mm, np.int32).view(np.uint32) ^ np.uint32(0x80000000)
    return ord_f(~k[:3]), ord_f(k[3:])


class _DeviceView:
    """zero-copy torch view of device memory the library owns (the filter's slots)"""

    def __init__(self, ptr, n, typestr):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False), "version": 2}


class FusedDevicePrims:
    """The four launches of the fused step on this rank's GPU (capi.PfFilter) with torch views of the slots the
    collectives move."""

    def __init__(self, ctx, field, capacity, d_scan_x, d_scan_y, n_points, decimation=3, std_frame=False):
        from . import capi
        self.capi, self.ctx = capi, ctx
        self.flt = capi.PfFilter(field, capacity)
        self.sx, self.sy, self.n_points, self.decim, self.std = d_scan_x, d_scan_y, n_points, decimation, std_frame
        dev = d_scan_x.device
        self.key = torch.as_tensor(_DeviceView(self.flt.d_key, 1, "<i8"), device=dev)
        self.rec = torch.as_tensor(_DeviceView(self.flt.d_record, 6, "<i8"), device=dev)
        self.mm = torch.as_tensor(_DeviceView(self.flt.d_minmax, 6, "<i4"), device=dev)
        self.n_local = 0

    def close(self):
        self.flt.close()

    def score_best(self, poses, first, predict=None, normals=None):
        self.n_local = poses.shape[0]
        self.flt.score(self.sx, self.sy, self.n_points, self.decim, poses, self.n_local, first, self.std, predict, normals)
        return self.key

    def weights(self, poses, keys):
        own = keys.data_ptr() == self.key.data_ptr()
        self.flt.weights(poses, None if own else keys, keys.shape[0])
        return self.rec

    def read(self, recs, n_total):
        """the records on the host: through the library's pinned mirror on the context's stream (device records), or parsed
        here (records a host-memory transport gathered)"""
        if not recs.is_cuda:
            return parse_records(recs.numpy(), n_total)
        own = recs.data_ptr() == self.rec.data_ptr()
        return self.flt.read(n_total, None if own else recs, 1 if own else recs.shape[0] // 6)

    def offspring(self, poses, offset, n_total, U0, W, m_first, m_count):
        out = torch.empty(m_count, 3, dtype=poses.dtype, device=poses.device)
        self.flt.offspring(offset, n_total, U0, W, m_first, m_count, poses, out)
        return out, self.mm

    def roughen(self, new, normals, mm, n_total, K):
        own = mm.data_ptr() == self.mm.data_ptr()
        self.flt.roughen(new.shape[0], n_total, K, normals, new, None if own else mm)


class FusedParticleFilter:
    """One localisation step over this rank's particle shard in four launches, one read-back and (several ranks) four
    collectives.  `prims`: score_best / weights / read / offspring / roughen (FusedDevicePrims, or stand-ins in the CPU
    tests).  With several ranks over RCCL the context has to launch on torch's current stream (`ctx.set_stream`): the
    collectives are ordered against that stream."""

    def __init__(self, n_total, prims, resample_threshold=0.5, roughen_K=0.0):
        self.rank, self.world = _world()
        self.n_total = n_total
        self.first, self.count = shard_bounds(n_total, self.rank, self.world)
        self.prims = prims
        self.resample_threshold = resample_threshold          # g_resample_threshold, globaldef.cpp:22
        self.roughen_K = roughen_K
        self.bytes_sent = 0
        self._stage = self.world > 1 and dist.get_backend() == "gloo"      # gloo moves host memory (CPU tests, rehearsals)

    def _all_gather(self, t):
        world = self.world
        if self._stage and t.is_cuda:
            h = t.cpu()
            out = torch.empty(world * h.shape[0], dtype=h.dtype)
            dist.all_gather(list(out.chunk(world)), h)
            return out.to(t.device)
        out = torch.empty(world * t.shape[0], dtype=t.dtype, device=t.device)
        if self._stage:
            dist.all_gather(list(out.chunk(world)), t)
        else:
            dist.all_gather_into_tensor(out, t)
        return out

    def step(self, poses, u0, predict=None, normals_predict=None, normals_roughen=None):
        """poses: this rank's [count, 3] tensor (updated in place by the prediction).  Returns (poses of this rank after
        the step, info dict)."""
        P, rank, world, N = self.prims, self.rank, self.world, self.n_total
        key = P.score_best(poses, self.first, predict, normals_predict)
        keys = self._all_gather(key) if world > 1 else key
        rec = P.weights(poses, keys)
        recs = self._all_gather(rec) if world > 1 else rec
        info, sums = P.read(recs, N)                                            # the step's read-back
        if world > 1:
            self.bytes_sent += (8 + RECORD.itemsize) * (world - 1)
        if not info["n_eff"] < self.resample_threshold * N:                     # particles.cpp:350
            return poses, info
        from . import capi
        W, U0, bounds = capi.pf_plan(sums, N, u0)
        p_lo, p_cnt = int(bounds[rank]), int(bounds[rank + 1] - bounds[rank])
        prod, mm = P.offspring(poses, sum(sums[:rank]), N, U0, W, p_lo, p_cnt)
        if world == 1:
            new = prod
        else:
            send, recv = exchange_plan(bounds, N, rank, world)
            new = torch.empty(self.count, 3, dtype=poses.dtype, device=poses.device)
            if self._stage and prod.is_cuda:
                out_h = torch.empty(self.count, 3, dtype=poses.dtype)
                dist.all_to_all_single(out_h, prod.cpu().contiguous(), output_split_sizes=recv, input_split_sizes=send)
                new.copy_(out_h)
                mm_h = mm.cpu()
                dist.all_reduce(mm_h, op=dist.ReduceOp.MAX)
                mm = mm_h.to(poses.device)
            else:
                dist.all_to_all_single(new, prod, output_split_sizes=recv, input_split_sizes=send)
                if self.roughen_K > 0:
                    dist.all_reduce(mm, op=dist.ReduceOp.MAX)
            self.bytes_sent += 12 * (sum(send) - send[rank]) + 24 * (world - 1)
        info["resampled"] = True
        if self.roughen_K > 0:
            P.roughen(new, normals_roughen, mm, N, self.roughen_K)
        return new, info
