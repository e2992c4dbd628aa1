"""CPU, world_size 2 and 3 (gloo): the exchange logic of the sharded particle-filter step (laser_slam_amd/pf.py,
SURVEY.md 8e) and of the sharded grid mapping, with numpy / oracle stand-ins for the per-rank HIP results, and the
N-rank launch path of bench.py.  The sharded result must equal the single-process result exactly."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _scores(poses):
    # any deterministic function of the pose stands in for lsa_score_poses
    return -((poses[:, 0] - 3.0) ** 2 + (poses[:, 1] + 1.0) ** 2) * 4.0 - torch.abs(poses[:, 2]) * 0.5


class NumpyPrims:
    """Stand-ins for the HIP kernels behind laser_slam_amd.pf.DevicePrims, in exact integer arithmetic (Python ints):
    the definitions csrc/shard_kernels.hip implements."""

    def score(self, p):
        return _scores(p)

    def argmax(self, s):
        m = float(s.max())
        return m, int(torch.nonzero(s == m)[0])

    def weights(self, s, gmax):
        w = torch.exp(s - gmax).float()
        wd = w.double()
        return w, float(wd.sum()), float((wd * wd).sum())

    @staticmethod
    def quantise(w):
        wd = np.clip(w.numpy().astype(np.float64), 0.0, 1.0)
        return [int(v) for v in np.floor(wd * 4294967296.0)]

    def fixed_scan(self, w):
        cum, c = [], 0
        for q in self.quantise(w):
            c += q
            cum.append(c)
        return cum, c

    def offspring_bounds(self, cum, offset, n_total, U0, W):
        e = []
        for c in cum:
            x = (offset + c) * n_total - U0
            e.append(0 if x <= 0 else min(n_total, -(-x // W)))
        return torch.tensor(e, dtype=torch.int32)

    def parents(self, e, m_first, m_count):
        m = np.arange(m_first, m_first + m_count)
        return torch.from_numpy(np.searchsorted(e.numpy(), m, side="right").astype(np.int32))

    def gather(self, poses, parent):
        return poses[parent.long()]


class NumpyFusedPrims:
    """Stand-ins for the four launches of the fused step (laser_slam_amd.pf.FusedDevicePrims, csrc/pfstep_kernels.hip) in
    Python integers: best key, weights / fixed-point cumulative sums / record, offspring by the definition, roughening."""

    def score_best(self, poses, first, predict=None, normals=None):
        from laser_slam_amd import pf
        if predict is not None:
            assert predict[0] == "predict"
            m, sg = np.asarray(predict[1], np.float64), np.asarray(predict[2], np.float64)
            poses.copy_(torch.from_numpy((m + normals.numpy().astype(np.float64) * sg).astype(np.float32)))
        self.s, self.first, self.poses = _scores(poses), first, poses
        best = max(pf.key_of(float(v), first + i) for i, v in enumerate(self.s))
        return torch.from_numpy(np.array([best], np.uint64).view(np.int64))

    def weights(self, poses, keys):
        from laser_slam_amd import pf
        gkey = max(int(k) for k in keys.numpy().view(np.uint64))
        gmax, gidx = pf.key_score_index(gkey)
        w = torch.exp(self.s - np.float32(gmax)).float()
        self.cum, c = [], 0
        for q in NumpyPrims.quantise(w):
            c += q
            self.cum.append(c)
        rec = np.zeros(1, pf.RECORD)
        wd = w.double()
        rec["q_sum"], rec["w_sum"], rec["w2_sum"], rec["best_key"] = c, float(wd.sum()), float((wd * wd).sum()), gkey
        if self.first <= gidx < self.first + len(self.s):
            rec["has_best"] = 1
            rec["best_pose"] = poses[gidx - self.first].numpy()
        return torch.from_numpy(rec.view(np.int64).copy())

    def read(self, recs, n_total):
        from laser_slam_amd import pf
        return pf.parse_records(recs.numpy(), n_total)

    def offspring(self, poses, offset, n_total, U0, W, m_first, m_count):
        from laser_slam_amd import pf
        parent, i = [], 0
        for m in range(m_first, m_first + m_count):
            while (offset + self.cum[i]) * n_total <= m * W + U0:
                i += 1
            parent.append(i)
        out = poses[torch.tensor(parent, dtype=torch.long)] if parent else poses[:0]
        return out.contiguous(), torch.from_numpy(pf.minmax_keys(out.numpy()))

    def roughen(self, new, normals, mm, n_total, K):
        from laser_slam_amd import pf
        lo, hi = pf.decode_minmax(mm.numpy())
        rootq = (hi.astype(np.float64) - lo.astype(np.float64)) * (K * float(n_total) ** (-1.0 / 3.0))
        new.copy_(torch.from_numpy((new.numpy().astype(np.float64) + normals.numpy().astype(np.float64) * rootq).astype(np.float32)))


def _reference_resample(w_all, n_total, u0):
    """Systematic resampling of the whole set straight from the definition: offspring m sits at m W + U0 on the axis of
    N-fold cumulative fixed-point weights; its parent is the first particle whose N-fold cumulative weight exceeds it."""
    q = NumpyPrims.quantise(w_all)
    W = sum(q)
    U0 = (int(np.ldexp(u0, 53)) * W) >> 53
    parent, i, c = [], 0, q[0]
    for m in range(n_total):
        while c * n_total <= m * W + U0:
            i += 1
            c += q[i]
        parent.append(i)
    return np.array(parent)


def _worker(rank, world, port, n_total, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from laser_slam_amd import pf
    rng = np.random.default_rng(5)
    poses_all = torch.from_numpy(rng.normal(0, 2, (n_total, 3)).astype(np.float32))
    f = pf.ShardedParticleFilter(n_total, NumpyPrims(), resample_threshold=0.9)
    mine = poses_all[f.first:f.first + f.count]
    new, info = f.step(mine, 0.37)
    assert new.shape[0] == f.count
    allnew = pf.gather_concat(new)
    sent = pf.gather_concat(torch.tensor([f.bytes_sent], dtype=torch.int64))
    if rank == 0:
        np.save(os.path.join(out_dir, "poses.npy"), allnew.numpy())
        np.save(os.path.join(out_dir, "sent.npy"), sent.numpy())
        np.save(os.path.join(out_dir, "info.npy"), np.array([info["best_score"], info["best_index"], info["weight_sum"],
                                                             info["n_eff"], float(info["resampled"])]))
    dist.barrier()
    dist.destroy_process_group()


def _fused_inputs(n_total):
    rng = np.random.default_rng(7)
    nrm_p = torch.from_numpy(rng.normal(0, 1, (n_total, 3)).astype(np.float32))
    nrm_r = torch.from_numpy(rng.normal(0, 1, (n_total, 3)).astype(np.float32))
    return nrm_p, nrm_r, ("predict", [3.0, -1.0, 0.1], [0.8, 0.7, 0.3])


def _fused_worker(rank, world, port, n_total, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from laser_slam_amd import pf
    nrm_p, nrm_r, predict = _fused_inputs(n_total)
    f = pf.FusedParticleFilter(n_total, NumpyFusedPrims(), resample_threshold=0.9, roughen_K=0.2)
    sl = slice(f.first, f.first + f.count)
    mine = torch.zeros(f.count, 3)
    new, info = f.step(mine, 0.37, predict=predict, normals_predict=nrm_p[sl].contiguous(), normals_roughen=nrm_r[sl].contiguous())
    assert new.shape[0] == f.count
    allnew = pf.gather_concat(new)
    sent = pf.gather_concat(torch.tensor([f.bytes_sent], dtype=torch.int64))
    if rank == 0:
        np.save(os.path.join(out_dir, "poses.npy"), allnew.numpy())
        np.save(os.path.join(out_dir, "sent.npy"), sent.numpy())
        np.save(os.path.join(out_dir, "info.npy"), np.array([info["best_score"], info["best_index"], info["weight_sum"], info["n_eff"],
                                                             float(info["resampled"])] + list(info["best_pose"])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_total", [(2, 1000), (3, 4099), (3, 1001)])
def test_fused_pf_step_equals_single_process(world, n_total, tmp_path):
    """The fused step (one key and one record all-gathered, surplus exchange, one all-reduce of the extremes) over 2 and 3
    gloo ranks gives exactly the single-process step - prediction, best particle, resampled set, roughening - and that is
    systematic resampling by its definition."""
    from laser_slam_amd import pf
    port = 29100 + (os.getpid() % 300) + 13 * world + (n_total % 5)
    mp.spawn(_fused_worker, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True)
    got, info, sent = np.load(tmp_path / "poses.npy"), np.load(tmp_path / "info.npy"), np.load(tmp_path / "sent.npy")
    nrm_p, nrm_r, predict = _fused_inputs(n_total)
    f = pf.FusedParticleFilter(n_total, NumpyFusedPrims(), resample_threshold=0.9, roughen_K=0.2)
    cur = torch.zeros(n_total, 3)
    ref, ref_info = f.step(cur, 0.37, predict=predict, normals_predict=nrm_p, normals_roughen=nrm_r)
    assert ref_info["resampled"] and info[4] == 1.0
    assert np.array_equal(got, ref.numpy())
    assert info[0] == ref_info["best_score"] and info[1] == ref_info["best_index"]
    assert np.array_equal(info[5:8].astype(np.float32), np.asarray(ref_info["best_pose"], np.float32))
    assert np.array_equal(cur.numpy()[ref_info["best_index"]], np.asarray(ref_info["best_pose"], np.float32))
    assert abs(info[2] - ref_info["weight_sum"]) < 1e-9 * ref_info["weight_sum"] and abs(info[3] - ref_info["n_eff"]) < 1e-9 * ref_info["n_eff"]
    # the definition: predicted poses -> weights -> parents -> roughening with the extremes of the new set
    w_all = torch.exp(_scores(cur) - np.float32(ref_info["best_score"])).float()
    parent = _reference_resample(w_all, n_total, 0.37)
    res = cur.numpy()[parent]
    rootq = (res.max(0).astype(np.float64) - res.min(0).astype(np.float64)) * (0.2 * float(n_total) ** (-1.0 / 3.0))
    assert np.array_equal(ref.numpy(), (res.astype(np.float64) + nrm_r.numpy().astype(np.float64) * rootq).astype(np.float32))
    per_rank = -(-n_total // world)
    assert sent.max() <= 12 * per_rank + 96 * world and sent.sum() > 0


def test_exchange_plan_host_arithmetic():
    """lsa_pf_exchange_plan (host side of the C ABI) against pf.exchange_plan for 2, 3 and 8 ranks and skewed weights:
    counts add up, own parts agree, what one rank sends to another is what that one expects from it."""
    from laser_slam_amd import capi, pf
    rng = np.random.default_rng(3)
    for world in (1, 2, 3, 8):
        for trial in range(6):
            n_total = int(rng.integers(world, 200000))
            sums = rng.integers(1, 1 << 40, world)
            if trial % 2:
                sums[rng.integers(0, world)] <<= 12          # one shard holds nearly all the weight
            if trial == 5 and world > 2:
                sums[1] = 0                                  # a shard without weight produces nothing
            W, U0, bounds = capi.pf_plan([int(v) for v in sums], n_total, float(rng.random()))
            plans = [capi.pf_exchange_plan(bounds, n_total, r, world) for r in range(world)]
            for r in range(world):
                send, recv = plans[r]
                assert (send, recv) == tuple(pf.exchange_plan(bounds, n_total, r, world))
                assert sum(send) == bounds[r + 1] - bounds[r] and sum(recv) == pf.shard_bounds(n_total, r, world)[1]
                assert send[r] == recv[r]
                for q in range(world):
                    assert send[q] == plans[q][1][r]


@pytest.mark.parametrize("world,n_total", [(2, 1000), (2, 1001), (3, 1000), (3, 4099)])
def test_sharded_pf_step_equals_single_process(world, n_total, tmp_path):
    """SURVEY 8(e)'s scheme: one 8-byte weight sum per rank is gathered, every rank resolves its own strata locally,
    only offspring that belong to another rank's block travel (12 B each).  The sharded step must give exactly the
    single-process step, and a rank must hand over O(N / world) bytes - the r1 scheme gathered 16 N bytes to every rank."""
    from laser_slam_amd import pf
    port = 29500 + (os.getpid() % 300) + 17 * world + (n_total % 7)
    mp.spawn(_worker, args=(world, port, n_total, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "poses.npy")
    info = np.load(tmp_path / "info.npy")
    sent = np.load(tmp_path / "sent.npy")
    # single-process run of the same step (world 1: no collective at all)
    rng = np.random.default_rng(5)
    poses_all = torch.from_numpy(rng.normal(0, 2, (n_total, 3)).astype(np.float32))
    f = pf.ShardedParticleFilter(n_total, NumpyPrims(), resample_threshold=0.9)
    ref, ref_info = f.step(poses_all, 0.37)
    assert ref_info["resampled"] and info[4] == 1.0
    assert got.shape == (n_total, 3) and np.array_equal(got, ref.numpy())
    assert info[1] == ref_info["best_index"] and info[0] == ref_info["best_score"]
    assert abs(info[2] - ref_info["weight_sum"]) < 1e-9 * ref_info["weight_sum"]
    assert abs(info[3] - ref_info["n_eff"]) < 1e-9 * ref_info["n_eff"]
    # ... which is systematic resampling by its definition
    w_all = NumpyPrims().weights(_scores(poses_all), ref_info["best_score"])[0]
    parent = _reference_resample(w_all, n_total, 0.37)
    assert np.array_equal(ref.numpy(), poses_all.numpy()[parent])
    # bytes handed to other ranks: the offspring outside the rank's own block + the records, not the particle set
    per_rank = -(-n_total // world)
    assert sent.max() <= 12 * per_rank + 64 * world, sent
    assert sent.max() < 16 * n_total / 4                      # (r1: 16 N per rank)
    assert sent.sum() > 0


def test_pf_plan_host_arithmetic():
    """lsa_pf_plan (host side of the C ABI, no GPU): total, U0 = floor(u0 W) and the offspring ranges of the shards
    against Python integers, incl. totals beyond 2^53 where doubles would no longer be exact."""
    from laser_slam_amd import capi
    rng = np.random.default_rng(11)
    for world in (1, 2, 3, 8):
        for scale in (1 << 20, 1 << 40, (1 << 59) // world):
            sums = [int(v) for v in rng.integers(1, scale, world)]
            n_total = int(rng.integers(world, 1 << 23))
            u0 = float(rng.random())
            W, U0, bounds = capi.pf_plan(sums, n_total, u0)
            assert W == sum(sums) and U0 == (int(np.ldexp(u0, 53)) * W) >> 53 and U0 < W
            off, want = 0, []
            for r in range(world + 1):
                x = off * n_total - U0
                want.append(0 if x <= 0 else min(n_total, -(-x // W)))
                if r < world:
                    off += sums[r]
            assert list(bounds) == want and bounds[0] == 0 and bounds[-1] == n_total
    with pytest.raises(capi.LsaError):
        capi.pf_plan([0, 0], 10, 0.5)                        # "total likelihood zero", SIRFlt.cpp:150-152
    with pytest.raises(capi.LsaError):
        capi.pf_plan([5, 5], 10, 1.0)


def test_shard_bounds_cover_everything():
    from laser_slam_amd import pf
    for n in (1, 7, 8, 1000, 1000003):
        for world in (1, 2, 3, 8):
            spans = [pf.shard_bounds(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (a, c), (b, _) in zip(spans, spans[1:]):
                assert a + c == b
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1


def _grid_inputs():
    rng = np.random.default_rng(9)
    S, nb = 37, 181
    rob = rng.uniform(-3, 3, (S, 2)).astype(np.float32)
    ang = rng.uniform(0, 2 * np.pi, (S, nb)); d = rng.uniform(0.2, 6, (S, nb))
    gx = (rob[:, :1] + d * np.cos(ang)).astype(np.float32); gy = (rob[:, 1:] + d * np.sin(ang)).astype(np.float32)
    return S, gx, gy, rob


def _grid_update_fn():
    from oracle_lib import OraclePMap
    S, gx, gy, rob = _grid_inputs()

    def update(first, count):
        o = OraclePMap(320, 320, 160, 160, 0.0, 0.0, 0.05)
        for k in range(first, first + count):
            o.update(gx[k], gy[k], float(rob[k, 0]), float(rob[k, 1]))
        hit, visits, _ = o.layers()
        return torch.from_numpy(hit.astype(np.int32)), torch.from_numpy(visits.astype(np.int32))

    return S, update


def _grid_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from laser_slam_amd import mapping
    S, update = _grid_update_fn()
    hit, visits = mapping.sharded_grid_update(S, update)
    if rank == 0:
        np.save(os.path.join(out_dir, "hit.npy"), hit.numpy()); np.save(os.path.join(out_dir, "sum.npy"), visits.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_grid_mapping_equals_single_process(tmp_path):
    """Config 5's exchange step: scans split over two ranks, count layers all-reduced."""
    port = 29950 + (os.getpid() % 40)
    mp.spawn(_grid_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    S, update = _grid_update_fn()
    hit, visits = update(0, S)
    assert np.array_equal(np.load(tmp_path / "hit.npy"), hit.numpy())
    assert np.array_equal(np.load(tmp_path / "sum.npy"), visits.numpy())
    assert int(hit.sum()) == 2 * S * 181 or int(hit.sum()) > 0


def test_bench_launches_n_ranks_itself():
    """`python bench.py --gpus N` without torchrun starts N ranks (child processes, before anything touches the GPU),
    they rendezvous, and rank 0 alone prints the line with n_gpus = N.  --rehearse-launch keeps the GPU out of it."""
    import json
    import subprocess
    for n in (1, 2, 3):
        res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1",
                              "--rehearse-launch"], capture_output=True, text=True, timeout=300,
                             env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")})
        assert res.returncode == 0, res.stderr[-2000:]
        lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1
        rec = json.loads(lines[0])
        assert rec["n_gpus"] == n and rec["steps"] == 2 and abs(rec["max_over_ranks_s"] - 0.001 * n) < 1e-12
    # under a launcher that sets the environment (torchrun), the process is one rank and does not spawn
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--rehearse-launch"], capture_output=True,
                         text=True, timeout=300, env=env)
    assert res.returncode == 0 and json.loads(res.stdout.strip().splitlines()[-1])["n_gpus"] == 1
    assert "WORLD_SIZE=1" in res.stderr
