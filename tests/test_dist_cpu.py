"""CPU, world_size 2 (gloo): the exchange logic of the sharded pose-scoring / particle-filter
step (laser_slam_amd/pf.py) and of the sharded scan-pair batches, with numpy stand-ins for the
per-rank HIP results.  The sharded result must equal the single-process result exactly."""
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


def _make_fns():
    from oracle_lib import Oracle  # systematic resampler restated in oracle/pose_oracle.c
    import ctypes as C
    L = Oracle("LMS211").L

    def score(p):
        return _scores(p)

    def argmax(s):
        m = float(s.max())
        return m, int(torch.nonzero(s == m)[0])

    def weights(s, gmax):
        w = torch.exp(s - gmax).float()
        wd = w.double()
        return w, float(wd.sum()), float((wd * wd).sum())

    def resample(all_w, u0, first, M):
        w = np.ascontiguousarray(all_w.numpy(), np.float32)
        n_total = len(w)
        parent = np.zeros(n_total, np.int32)
        L.orc_systematic_resample(w.ctypes.data_as(C.c_void_p), n_total, C.c_double(u0), n_total,
                                  parent.ctypes.data_as(C.c_void_p))
        return torch.from_numpy(parent[first:first + M].copy())

    def gather(all_p, parent):
        return all_p[parent.long()]

    return score, argmax, weights, resample, gather


def _worker(rank, world, port, n_total, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from laser_slam_amd import pf
    rng = np.random.default_rng(5)
    poses_all = torch.from_numpy(rng.normal(0, 2, (n_total, 3)).astype(np.float32))
    f = pf.ShardedParticleFilter(n_total, *_make_fns(), resample_threshold=0.9)
    mine = poses_all[f.first:f.first + f.count]
    new, info = f.step(mine, 0.37)
    allnew = pf.gather_concat(new)
    if rank == 0:
        np.save(os.path.join(out_dir, "poses.npy"), allnew.numpy())
        np.save(os.path.join(out_dir, "info.npy"), np.array([info["best_score"], info["best_index"], info["weight_sum"],
                                                             info["n_eff"], float(info["resampled"])]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [1000, 1001])
def test_sharded_pf_step_equals_single_process(n_total, tmp_path):
    from laser_slam_amd import pf
    port = 29500 + (os.getpid() % 400) + (n_total % 7)
    mp.spawn(_worker, args=(2, port, n_total, str(tmp_path)), nprocs=2, join=True)
    got = np.load(tmp_path / "poses.npy")
    info = np.load(tmp_path / "info.npy")
    # single-process reference of the same step
    rng = np.random.default_rng(5)
    poses_all = torch.from_numpy(rng.normal(0, 2, (n_total, 3)).astype(np.float32))
    f = pf.ShardedParticleFilter(n_total, *_make_fns(), resample_threshold=0.9)
    ref, ref_info = f.step(poses_all, 0.37)
    assert ref_info["resampled"] and info[4] == 1.0
    assert np.array_equal(got, ref.numpy())
    assert info[1] == ref_info["best_index"] and info[0] == ref_info["best_score"]
    assert abs(info[2] - ref_info["weight_sum"]) < 1e-9 * ref_info["weight_sum"]
    assert abs(info[3] - ref_info["n_eff"]) < 1e-9 * ref_info["n_eff"]


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
