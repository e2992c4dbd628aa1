"""GPU: the sharded particle-filter primitives (csrc/shard_kernels.hip) and the C ABI's communicator entry points.

One GPU is all a test box has, so (a) the fixed-point primitives are run for several VIRTUAL shards of one particle set
on the one device and stitched together with the host plan - the result must be the single-shard result and the
definition in Python integers, which is what makes the multi-GPU step independent of the split; (b) the RCCL path of
`lsa_comm` is exercised with a real one-rank communicator (LSA_COMM_FORCE_RCCL=1).  The multi-rank exchange logic
itself is covered on CPU ranks by tests/test_dist_cpu.py."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _py_quantise(w):
    return [int(v) for v in np.floor(np.clip(w.astype(np.float64), 0.0, 1.0) * 4294967296.0)]


def _py_bounds(q, offset, n_total, U0, W):
    e, c = [], offset
    for v in q:
        c += v
        x = c * n_total - U0
        e.append(0 if x <= 0 else min(n_total, -(-x // W)))
    return np.array(e, np.int64)


@pytest.mark.parametrize("n_total,sharp", [(5000, 1.0), (100003, 6.0), (1 << 20, 30.0)])
def test_fixed_point_resampling_is_split_independent(n_total, sharp):
    import torch
    from laser_slam_amd import capi, pf
    ctx = capi.Context(capi.param_preset("LMS211"))
    rng = np.random.default_rng(n_total)
    w = np.exp(-sharp * rng.random(n_total) ** 2).astype(np.float32)
    w[rng.integers(0, n_total, 5)] = 1.0
    w[rng.integers(0, n_total, 50)] = 0.0                     # particles that must never be drawn
    poses = rng.normal(0, 3, (n_total, 3)).astype(np.float32)
    u0 = 0.6180339887
    q = _py_quantise(w)
    d_w, d_p = torch.from_numpy(w).cuda(), torch.from_numpy(poses).cuda()
    torch.cuda.synchronize()
    # the whole set as one shard
    d_cum = torch.zeros(n_total, dtype=torch.int64, device="cuda")
    assert capi.pf_fixed_scan(ctx, d_w, n_total, d_cum) == sum(q)
    assert np.array_equal(d_cum.cpu().numpy(), np.cumsum(np.array(q, np.int64)))
    W, U0, bounds = capi.pf_plan([sum(q)], n_total, u0)
    d_e = torch.zeros(n_total, dtype=torch.int32, device="cuda")
    capi.pf_offspring_bounds(ctx, d_cum, n_total, 0, n_total, U0, W, d_e); ctx.sync()
    e_one = d_e.cpu().numpy()
    if n_total <= 200000:
        assert np.array_equal(e_one, _py_bounds(q, 0, n_total, U0, W))
    assert e_one[-1] == n_total and np.all(np.diff(e_one) >= 0)
    d_par = torch.zeros(n_total, dtype=torch.int32, device="cuda")
    capi.pf_parents(ctx, d_e, n_total, 0, n_total, d_par); ctx.sync()
    par_one = d_par.cpu().numpy()
    assert np.array_equal(par_one, np.searchsorted(e_one, np.arange(n_total), side="right"))
    assert np.all(w[par_one] > 0)
    # offspring counts follow the weights: N w / sum(w) within one
    cnt = np.bincount(par_one, minlength=n_total)
    expect = n_total * np.array(q, np.float64) / W
    assert np.all(np.abs(cnt - expect) < 1.0 + 1e-9)
    # the same set as 3 and as 7 virtual shards: stitched offspring == the one-shard result
    for world in (3, 7):
        sums, blocks, cums = [], [], []
        for r in range(world):
            f, c = pf.shard_bounds(n_total, r, world)
            assert (f, c) == capi.shard_bounds(n_total, r, world)
            blocks.append((f, c))
            cums.append(torch.zeros(c, dtype=torch.int64, device="cuda"))
            sums.append(capi.pf_fixed_scan(ctx, d_w[f:f + c], c, cums[r]))
        assert sums == [sum(q[f:f + c]) for f, c in blocks]
        W2, U02, b2 = capi.pf_plan(sums, n_total, u0)
        assert (W2, U02) == (W, U0) and b2[0] == 0 and b2[-1] == n_total
        new = np.zeros((n_total, 3), np.float32)
        sent = []
        for r, (f, c) in enumerate(blocks):
            lo, cnt_r = int(b2[r]), int(b2[r + 1] - b2[r])
            assert lo == (e_one[f - 1] if f > 0 else 0)
            if cnt_r == 0:
                sent.append(0)
                continue
            d_er = torch.zeros(c, dtype=torch.int32, device="cuda")
            capi.pf_offspring_bounds(ctx, cums[r], c, sum(sums[:r]), n_total, U0, W, d_er)
            d_pr = torch.zeros(cnt_r, dtype=torch.int32, device="cuda")
            capi.pf_parents(ctx, d_er, c, lo, cnt_r, d_pr)
            d_out = torch.zeros(cnt_r, 3, device="cuda")
            capi.gather_poses(ctx, d_p[f:f + c], d_pr, cnt_r, d_out); ctx.sync()
            assert np.array_equal(d_er.cpu().numpy(), e_one[f:f + c])
            new[lo:lo + cnt_r] = d_out.cpu().numpy()
            send, recv = pf.exchange_plan(b2, n_total, r, world)
            assert sum(send) == cnt_r and sum(recv) == c
            sent.append(12 * (sum(send) - send[r]))
        assert np.array_equal(new, poses[par_one])
        assert max(sent) <= 12 * n_total                     # (a shard holding all the weight sends at most the set once)
    ctx.close()


def test_comm_one_rank_over_rccl(monkeypatch):
    """lsa_comm with a real RCCL communicator of one rank: ncclCommInitRank, all-gather, all-reduce and grouped
    send / recv run on the device; results equal the single-GPU entry points."""
    import torch
    from laser_slam_amd import capi
    monkeypatch.setenv("LSA_COMM_FORCE_RCCL", "1")
    ctx = capi.Context(capi.param_preset("LMS211"))
    comm = capi.Comm(ctx, 0, 1, capi.comm_unique_id())
    n = 70001
    rng = np.random.default_rng(3)
    score = (-4.0 * rng.random(n)).astype(np.float32)
    score[4242] = 0.5; score[60000] = 0.5                      # a tie: the smaller index wins
    poses = rng.normal(0, 2, (n, 3)).astype(np.float32)
    d_s, d_p = torch.from_numpy(score).cuda(), torch.from_numpy(poses).cuda()
    d_w = torch.zeros(n, device="cuda")
    torch.cuda.synchronize()
    sc, idx, pose = comm.pf_global_best(d_s, d_p, n, 1000)
    assert sc == np.float32(0.5) and idx == 1000 + 4242 and np.array_equal(pose, poses[4242])
    s1, s2 = comm.pf_global_weights(d_s, n, sc, d_w)
    w = np.exp(score - np.float32(0.5)).astype(np.float32)
    assert np.allclose(d_w.cpu().numpy(), w, rtol=2e-6)
    assert abs(s1 - w.astype(np.float64).sum()) < 1e-5 * s1 and abs(s2 - (w.astype(np.float64) ** 2).sum()) < 1e-5 * s2
    d_new = torch.zeros(n, 3, device="cuda")
    comm.pf_resample_sharded(d_w, d_p, n, n, 0.25, d_new)
    # against the primitives
    wd = d_w.cpu().numpy()
    q = _py_quantise(wd)
    W, U0, _ = capi.pf_plan([sum(q)], n, 0.25)
    par = np.searchsorted(_py_bounds(q, 0, n, U0, W), np.arange(n), side="right")
    assert np.array_equal(d_new.cpu().numpy(), poses[par])
    # roughening with the whole set's extremes == the single-GPU call
    nrm = rng.normal(0, 1, (n, 3)).astype(np.float32)
    d_n = torch.from_numpy(nrm).cuda()
    a, b = d_new.clone(), d_new.clone()
    torch.cuda.synchronize()
    comm.pf_roughen_sharded(n, n, 0.2, d_n, a)
    capi.pf_roughen(ctx, n, 0.2, d_n, b); ctx.sync()
    assert np.array_equal(a.cpu().numpy(), b.cpu().numpy())
    # grid all-reduce of one rank leaves the counts as they are
    grid = capi.Grid(ctx, 256, 256, 128, 128, 0.0, 0.0, 0.05)
    gx = (rng.random(181) * 4 - 2).astype(np.float32); gy = (rng.random(181) * 4 - 2).astype(np.float32)
    grid.update_host(gx, gy, 0.1, -0.2)
    before = [grid.download(0).copy(), grid.download(1).copy()]
    comm.grid_allreduce(grid); ctx.sync()
    assert np.array_equal(grid.download(0), before[0]) and np.array_equal(grid.download(1), before[1])
    assert before[0].sum() > 0
    grid.close(); comm.close(); ctx.close()


def _two_rank_inputs(n_total):
    rng = np.random.default_rng(21)
    sx, sy, res = 260, 200, 0.05
    prob = np.full((sx, sy), 0.1, np.float32)
    prob[:, :3] = 0.9; prob[:, -3:] = 0.9; prob[:3, :] = 0.9; prob[-3:, :] = 0.9; prob[100:104, 60:140] = 0.95
    n = 361
    ang = np.linspace(-np.pi / 2, np.pi / 2, n)
    scan_x = (3.0 * np.cos(ang)).astype(np.float32); scan_y = (3.0 * np.sin(ang)).astype(np.float32)
    nrm_p = rng.normal(0, 1, (n_total, 3)).astype(np.float32); nrm_r = rng.normal(0, 1, (n_total, 3)).astype(np.float32)
    return prob, (sx, sy, res), n, scan_x, scan_y, nrm_p, nrm_r, ("predict", [6.5, 5.0, 0.2], [2.0, 1.5, 1.0])


def _two_rank_step(rank, world, n_total, out_dir):
    import torch
    from laser_slam_amd import capi, pf
    prob, (sx, sy, res), n, scan_x, scan_y, nrm_p, nrm_r, predict = _two_rank_inputs(n_total)
    ctx = capi.Context(capi.param_preset("LMS211"))
    stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); ctx.set_stream(stream.cuda_stream)
    fld = capi.Field(ctx, sx, sy, 0, 0, 0.0, 0.0, res)
    fld.build(torch.from_numpy(prob).cuda(), 1.0, 0.3, 0.5)
    first, cnt = pf.shard_bounds(n_total, rank, world)
    d_x, d_y = torch.from_numpy(scan_x).cuda(), torch.from_numpy(scan_y).cuda()
    fp = pf.FusedDevicePrims(ctx, fld, cnt, d_x, d_y, n, 3)
    f = pf.FusedParticleFilter(n_total, fp, resample_threshold=2.0, roughen_K=0.2)
    cur = torch.zeros(cnt, 3, device="cuda")
    d_np = torch.from_numpy(nrm_p[first:first + cnt].copy()).cuda(); d_nr = torch.from_numpy(nrm_r[first:first + cnt].copy()).cuda()
    new, info = f.step(cur, 0.37, predict=predict, normals_predict=d_np, normals_roughen=d_nr)
    torch.cuda.synchronize()
    allnew = pf.gather_concat(new.cpu())
    if rank == 0:
        np.save(os.path.join(out_dir, "poses_w%d.npy" % world), allnew.numpy())
        np.save(os.path.join(out_dir, "info_w%d.npy" % world), np.array([info["best_score"], info["best_index"], info["n_eff"], f.bytes_sent]))
    fp.close(); fld.close(); ctx.close()


def _two_rank_worker(rank, world, port, n_total, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _two_rank_step(rank, world, n_total, out_dir)
    dist.barrier()
    dist.destroy_process_group()


def test_fused_step_two_ranks_on_one_gpu_equal_one_rank(tmp_path):
    """The fused step's exchange with real device buffers: two processes (gloo, both on this box's one GPU; the collectives
    move host copies) run pf.FusedParticleFilter over the two halves of 100 001 particles - best keys and records gathered,
    the offspring that change halves exchanged, the extremes reduced - and the concatenated result equals the step of one
    process over the whole set bit for bit.  What RCCL would carry instead of gloo are the same tensors."""
    import torch.multiprocessing as mp
    n_total = 100001
    port = 29650 + (os.getpid() % 200)
    mp.spawn(_two_rank_worker, args=(2, port, n_total, str(tmp_path)), nprocs=2, join=True)
    _two_rank_step(0, 1, n_total, str(tmp_path))
    two, one = np.load(tmp_path / "poses_w2.npy"), np.load(tmp_path / "poses_w1.npy")
    i2, i1 = np.load(tmp_path / "info_w2.npy"), np.load(tmp_path / "info_w1.npy")
    assert two.shape == (n_total, 3) and np.array_equal(two, one)
    assert i2[0] == i1[0] and i2[1] == i1[1] and abs(i2[2] - i1[2]) < 1e-9 * i1[2]
    assert i1[3] == 0 and 0 < i2[3] < 12 * n_total / 4
