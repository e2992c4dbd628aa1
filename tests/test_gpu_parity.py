"""GPU parity: the HIP path (through the C ABI) against the golden vectors and the
CPU oracle on identical inputs.  Bars: bit-exact for flags / segments / indices /
grid counts; poses within 1e-4 m (= 1e-2 cm in the matcher's cm units) and
1e-4 rad as north_star states — in practice the kernels reproduce the reference
bit for bit and the tests report how many pairs are bit-identical."""
import os

import numpy as np
import pytest

from helpers import GOLDEN, SETS, golden_param, load, same_bits

pytestmark = pytest.mark.gpu

TOL_CM = 1e-2     # 1e-4 m
TOL_RAD = 1e-4


@pytest.fixture(scope="module")
def torch_mod():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return torch


def make_ctx(name, g):
    from laser_slam_amd import capi, synth
    laser = SETS[name][0]
    p = capi.param_preset(laser) if laser else capi.param_make(**{k: v for k, v in synth.SYN1081.items()})
    gp = golden_param(g)
    p.fi_min = float(np.float32(gp["fi_min"]))
    p.fi_max = float(np.float32(gp["fi_max"]))
    p.max_range = int(gp["max_range"])
    return capi.Context(p), p


def dev(torch, a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a).astype(dtype)).cuda()


def run_match(torch, ctx, which, r, bad, seg, ref_idx, act_idx, init=None, want_err=True):
    from laser_slam_amd import capi  # noqa: F401
    B = len(ref_idx)
    d_r = dev(torch, r, np.float32); d_bad = dev(torch, bad, np.uint8); d_seg = dev(torch, seg, np.int16)
    d_ri = dev(torch, ref_idx, np.int32); d_ai = dev(torch, act_idx, np.int32)
    d_init = None if init is None else dev(torch, init, np.float32)
    pose = torch.zeros(B, 3, dtype=torch.float32, device="cuda")
    ret = torch.zeros(B, dtype=torch.float32, device="cuda")
    exy = torch.zeros(B, 2, dtype=torch.float32, device="cuda") if want_err else None
    na = torch.zeros(B, dtype=torch.int32, device="cuda") if want_err else None
    st = torch.zeros(B, dtype=torch.int32, device="cuda")
    it = torch.zeros(B, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()          # the context launches on its own stream: fills must have landed
    ctx.match_batch(which, B, d_r, d_bad, d_seg, d_ri, d_ai, pose, d_init, ret, exy, na, st, it)
    ctx.sync()
    out = dict(pose=pose.cpu().numpy(), ret=ret.cpu().numpy(), status=st.cpu().numpy(), iters=it.cpu().numpy())
    if want_err:
        out.update(exy=exy.cpu().numpy(), nassoc=na.cpu().numpy())
    return out


def pose_close(a, b):
    return (np.abs(a[:, :2] - b[:, :2]).max(axis=1) <= TOL_CM) & (np.abs(a[:, 2] - b[:, 2]) <= TOL_RAD)


@pytest.mark.parametrize("name", list(SETS))
def test_tables_match_reference(name, torch_mod):
    from laser_slam_amd import capi
    g = load(name)
    ctx, p = make_ctx(name, g)
    if "tab_fi" in g:
        fi, si, co = capi.param_tables(p)
        assert same_bits(fi, g["tab_fi"]) and same_bits(si, g["tab_si"]) and same_bits(co, g["tab_co"])
    ctx.close()


@pytest.mark.parametrize("path", ["wave", "walk", "lanes", "tiles"])
@pytest.mark.parametrize("name", list(SETS))
def test_preprocess(name, path, torch_mod, monkeypatch):
    """pm_preprocessScan against the golden vectors through the four kernels: one wave per scan with the whole wave at
    work (small batches, the host entry point) or with lane 0 walking the scan the literal way ("walk"), one lane per scan
    through LDS tiles (batches of 4096 scans and more; forced here by LSA_PREPROCESS_PATH) and its untiled predecessor."""
    monkeypatch.setenv("LSA_PREPROCESS_PATH", path)
    torch = torch_mod
    g = load(name)
    ctx, _ = make_ctx(name, g)
    S = g["raw_r"].shape[0]
    d_r = dev(torch, g["raw_r"], np.float32); d_bad = dev(torch, g["raw_bad"], np.uint8)
    d_seg = torch.zeros(g["raw_r"].shape, dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    ctx.preprocess_batch(S, d_r, d_bad, d_seg)
    ctx.sync()
    assert same_bits(d_r.cpu().numpy(), g["pre_r"])
    assert np.array_equal(d_bad.cpu().numpy().astype(np.int32), g["pre_bad"])
    assert np.array_equal(d_seg.cpu().numpy().astype(np.int32), g["pre_seg"])
    # host single-scan entry point
    r, bad, seg = ctx.preprocess_host(g["raw_r"][1], g["raw_bad"][1])
    assert same_bits(r, g["pre_r"][1]) and np.array_equal(bad, g["pre_bad"][1]) and np.array_equal(seg, g["pre_seg"][1])
    ctx.close()


@pytest.mark.parametrize("points,scans", [(1081, 4999), (361, 777), (181, 64 * 9 + 1), (97, 130), (8, 70)])
def test_preprocess_paths_agree_on_ragged_batches(points, scans, torch_mod, monkeypatch):
    """The three preprocessing kernels on batches that are not a multiple of a wave's 64 scans, on five sensor sizes (the
    tiled kernel moves 16-beam tiles: 97 and 8 beams end inside a tile), with spikes, far readings and flagged beams: the
    other kernels must return the bits of the literal walk, which the golden vectors pin."""
    torch = torch_mod
    from laser_slam_amd import capi, synth
    rng = np.random.default_rng(points)
    par = dict(synth.SYN1081, points=points, window=min(synth.SYN1081["window"], (points - 1) // 2))
    ctx = capi.Context(capi.param_make(**par), 0)
    base = (300.0 + 200.0 * np.sin(np.arange(points) * 0.05)[None, :] + rng.normal(0, 2.0, (scans, points))).astype(np.float32)
    base += rng.normal(0, 40.0, base.shape).astype(np.float32) * (rng.random(base.shape) < 0.05)
    base[rng.random(base.shape) < 0.01] = 9000.0
    flags = (rng.random(base.shape) < 0.03).astype(np.uint8)
    outs = {}
    for path in ("walk", "wave", "lanes", "tiles"):
        monkeypatch.setenv("LSA_PREPROCESS_PATH", path)
        d_r = dev(torch, base, np.float32); d_bad = dev(torch, flags, np.uint8)
        d_seg = torch.zeros(base.shape, dtype=torch.int16, device="cuda")
        torch.cuda.synchronize()
        ctx.preprocess_batch(scans, d_r, d_bad, d_seg)
        ctx.sync()
        outs[path] = (d_r.cpu().numpy().view(np.uint32), d_bad.cpu().numpy(), d_seg.cpu().numpy())
    assert outs["walk"][2].max() > 1 and outs["walk"][1].sum() > flags.sum()
    for path in ("wave", "lanes", "tiles"):
        for a, b in zip(outs["walk"], outs[path]):
            assert np.array_equal(a, b), path
    ctx.close()


@pytest.mark.parametrize("name", list(SETS))
def test_scan_project(name, torch_mod):
    torch = torch_mod
    g = load(name)
    ctx, _ = make_ctx(name, g)
    idx = g["stage_idx"]
    d_r = dev(torch, g["pre_r"][idx], np.float32); d_bad = dev(torch, g["pre_bad"][idx], np.uint8)
    d_seg = dev(torch, g["pre_seg"][idx], np.int16); d_pose = dev(torch, g["stage_pose"], np.float32)
    nr = torch.zeros(d_r.shape, dtype=torch.float32, device="cuda")
    nb = torch.zeros(d_r.shape, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ctx.scan_project_batch(len(idx), d_r, d_bad, d_seg, d_pose, nr, nb)
    ctx.sync()
    assert same_bits(nr.cpu().numpy(), g["stage_new_r"])
    assert np.array_equal(nb.cpu().numpy().astype(np.int32), g["stage_new_bad"])
    ctx.close()


@pytest.mark.parametrize("name", list(SETS))
def test_psm_stages_host_api(name, torch_mod):
    """pm_scan_project / pm_orientation_search / pm_translation_estimation / single
    preprocess stages through the host entry points the drop-in classes use."""
    g = load(name)
    ctx, _ = make_ctx(name, g)
    for t, k in enumerate(g["stage_idx"]):
        act = (g["pre_r"][k], g["pre_bad"][k], g["pre_seg"][k])
        nr, nb = ctx.scan_project_host(*act, g["stage_pose"][t])
        assert same_bits(nr, g["stage_new_r"][t]) and np.array_equal(nb, g["stage_new_bad"][t])
        st, dth = ctx.orientation_search_host(g["pre_r"][k - 1], g["pre_bad"][k - 1], nr, nb)
        assert (st != 0) == (g["stage_orient"][t, 0] != 0)
        if st == 0:
            assert np.float32(dth) == g["stage_orient"][t, 1]
        st, dx, dy, err = ctx.translation_estimation_host(g["pre_r"][k - 1], g["pre_bad"][k - 1], nr, nb)
        assert (st != 0) == (g["stage_trans"][t, 0] != 0)
        if st == 0:
            assert same_bits(np.array([dx, dy, err], np.float32), g["stage_trans"][t, 1:])
    n = g["raw_r"].shape[1]
    z = np.zeros(n, np.int32)
    for k in range(g["median_only"].shape[0]):
        r, bad, seg = ctx.preprocess_stages_host(g["raw_r"][k], g["raw_bad"][k], z, 1)      # median only
        assert same_bits(r, g["median_only"][k]) and np.array_equal(bad, g["raw_bad"][k])
        r2, bad2, seg2 = ctx.preprocess_stages_host(r, bad, z, 2)                            # far only
        assert same_bits(r2, r) and np.array_equal(bad2, g["pre_bad"][k])
        r3, bad3, seg3 = ctx.preprocess_stages_host(r2, bad2, z, 4)                          # segment only
        assert np.array_equal(seg3, g["pre_seg"][k]) and np.array_equal(bad3, bad2)
    ctx.close()


@pytest.mark.parametrize("name", list(SETS))
@pytest.mark.parametrize("which", [0, 1])
def test_match_vs_golden(name, which, torch_mod):
    torch = torch_mod
    g = load(name)
    ctx, _ = make_ctx(name, g)
    nm = "psm" if which == 0 else "icp"
    S = g["pre_r"].shape[0]
    ref_idx = np.arange(S - 1); act_idx = ref_idx + 1
    out = run_match(torch, ctx, which, g["pre_r"], g["pre_bad"], g["pre_seg"], ref_idx, act_idx)
    assert np.array_equal(out["status"] != 0, g[nm + "_status"] != 0)
    ok = g[nm + "_status"] == 0
    close = pose_close(out["pose"], g[nm + "_pose"])
    exact = np.array([same_bits(out["pose"][b], g[nm + "_pose"][b]) for b in range(S - 1)])
    print("%s %s: %d/%d pairs bit-identical, %d within tolerance" % (name, nm, exact.sum(), S - 1, close.sum()))
    assert close.all(), np.nonzero(~close)[0]
    assert exact.all(), np.nonzero(~exact)[0]          # the design target is bit-faithful FP32
    assert same_bits(out["ret"][ok], g[nm + "_ret"][ok])
    assert np.array_equal(out["nassoc"][ok], g[nm + "_nassoc"][ok])
    has = ok & (g[nm + "_nassoc"] > 0)
    assert same_bits(out["exy"][has], g[nm + "_exy"][has])
    # failed pairs report (-1,-1) like FMatch
    assert np.all(out["exy"][~ok] == -1.0)
    # non-zero initial poses
    K = g["init_pose"].shape[0]
    out = run_match(torch, ctx, which, g["pre_r"], g["pre_bad"], g["pre_seg"], ref_idx[:K], act_idx[:K],
                    init=g["init_pose"], want_err=False)
    assert np.array_equal(out["status"] != 0, g[nm + "_init_status"] != 0)
    assert same_bits(out["pose"], g[nm + "_init_pose_out"])
    ctx.close()


@pytest.mark.parametrize("name", ["mit", "syn1081"])
def test_match_vs_oracle_shuffled_pairs(name, torch_mod):
    """Pairs that are NOT consecutive (harder, more failures) against the CPU oracle."""
    torch = torch_mod
    from oracle_lib import SYN1081, Oracle
    g = load(name)
    ctx, p = make_ctx(name, g)
    laser = SETS[name][0]
    o = Oracle(laser) if laser else Oracle(**SYN1081)
    o.set_fimin_maxrange(np.float32(p.fi_min), p.max_range)
    S = g["pre_r"].shape[0]
    rng = np.random.default_rng(7)
    B = 48
    ref_idx = rng.integers(0, S, B); act_idx = np.clip(ref_idx + rng.integers(-4, 5, B), 0, S - 1)
    r = np.ascontiguousarray(g["pre_r"]); bad = np.ascontiguousarray(g["pre_bad"]); seg = np.ascontiguousarray(g["pre_seg"])
    for which in (0, 1):
        out = run_match(torch, ctx, which, r, bad, seg, ref_idx, act_idx, want_err=False)
        pose_o, st_o = o.match_batch(which, r, bad, seg, ref_idx, act_idx)
        assert np.array_equal(out["status"], st_o)
        assert same_bits(out["pose"], pose_o)
    ctx.close()


@pytest.mark.parametrize("name", ["mit", "fr079", "intel", "indoor"])
def test_match_wide_pairs_vs_oracle(name, torch_mod):
    """Every pair of the log head up to 6 scans apart (rotations of tens of degrees: bearings far outside the
    search window, empty candidate ranges, many failures), a third with random priors, both matchers,
    against the CPU oracle.  (This is the test that caught an empty-range bug in the straight-line NN path.)"""
    torch = torch_mod
    from oracle_lib import Oracle
    g = load(name)
    ctx, p = make_ctx(name, g)
    o = Oracle(SETS[name][0]); o.set_fimin_maxrange(np.float32(p.fi_min), p.max_range)
    r, bad, seg = (np.ascontiguousarray(g[k]) for k in ("pre_r", "pre_bad", "pre_seg"))
    S = r.shape[0]
    pairs = [(i, j) for i in range(S) for j in range(max(0, i - 6), min(S, i + 7))]
    ref_idx = np.array([a for a, _ in pairs], np.int32); act_idx = np.array([b for _, b in pairs], np.int32)
    B = len(pairs)
    rng = np.random.default_rng(3)
    init = np.zeros((B, 6), np.float32)
    init[:, 0:2] = rng.uniform(-200, 200, (B, 2)); init[:, 2] = rng.uniform(-3, 3, B)
    init[:, 3:5] = init[:, 0:2] + rng.uniform(-40, 40, (B, 2)); init[:, 5] = init[:, 2] + rng.uniform(-0.15, 0.15, B)
    init[np.arange(B) % 3 != 1] = 0
    for which in (0, 1):
        out = run_match(torch, ctx, which, r, bad, seg, ref_idx, act_idx, init=init, want_err=False)
        mism = 0
        for k in range(B):
            st, pose, _, _ = o.match(which, (r[ref_idx[k]], bad[ref_idx[k]], seg[ref_idx[k]]),
                                     (r[act_idx[k]], bad[act_idx[k]], seg[act_idx[k]]), init[k, :3], init[k, 3:])
            mism += (st != out["status"][k]) or (st == 0 and not same_bits(pose, out["pose"][k]))
        assert mism == 0, (name, which, mism)
    ctx.close()


def test_match_soak_vs_oracle(torch_mod):
    """A wider net than the golden sets: 900 pairs of synthetic 1081-beam scans (several noise seeds; scans
    0-3 apart, both directions) with random initial poses, both matchers, against the CPU oracle: every
    status equal and every pose bit-identical."""
    torch = torch_mod
    from laser_slam_amd import capi, synth
    from oracle_lib import SYN1081, Oracle
    rng = np.random.default_rng(2024)
    raws = [synth.make_scans(40, noise_seed=synth.WORLD_SEED + 100 + j)[0] for j in range(3)]
    raw = np.concatenate(raws, 0)
    o = Oracle(**SYN1081)
    ctx = capi.Context(capi.param_make(**synth.SYN1081))
    pre = [o.preprocess(raw[k]) for k in range(len(raw))]
    r = np.stack([q[0] for q in pre]); bad = np.stack([q[1] for q in pre]); seg = np.stack([q[2] for q in pre])
    B = 450
    base = rng.integers(0, 3, B) * 40
    a = rng.integers(0, 40, B); b = np.clip(a + rng.integers(-3, 4, B), 0, 39)
    ref_idx, act_idx = base + a, base + b
    # init = (ref pose, act pose) as the callers pass them: a random world pose and a perturbed relative guess
    init = np.zeros((B, 6), np.float32)
    init[:, 0:2] = rng.uniform(-300, 300, (B, 2)); init[:, 2] = rng.uniform(-3, 3, B)
    init[:, 3:5] = init[:, 0:2] + rng.uniform(-25, 25, (B, 2)); init[:, 5] = init[:, 2] + rng.uniform(-0.05, 0.05, B)
    init[: B // 3] = 0                                     # a third with the zero guess of the front end
    bad_pairs = 0
    for which in (0, 1):
        out = run_match(torch, ctx, which, r, bad, seg, ref_idx, act_idx, init=init, want_err=False)
        for k in range(B):
            st, pose, _, _ = o.match(which, pre[ref_idx[k]], pre[act_idx[k]], init[k, :3], init[k, 3:])
            if st != out["status"][k] or (st == 0 and not same_bits(pose, out["pose"][k])):
                bad_pairs += 1
    assert bad_pairs == 0
    ctx.close()


@pytest.mark.parametrize("n_beams,window", [(64, 20), (200, 40), (777, 150), (1000, 200), (1280, 200)])
def test_match_vs_oracle_other_beam_counts(n_beams, window, torch_mod):
    """Beam counts outside the presets run the size-generic kernel (the presets' sizes are compiled
    with the beam count as a constant): synthetic scans of odd sizes against the CPU oracle,
    preprocessing included."""
    torch = torch_mod
    from laser_slam_amd import capi, synth
    from oracle_lib import Oracle
    kw = dict(points=n_beams, fov_deg=270, max_range=3000, min_valid=max(8, n_beams // 4), window=window,
              fi_min_deg=-135, fi_max_deg=135)
    raw, _ = synth.make_scans(10, n_beams=n_beams, noise_seed=4242 + n_beams)
    o = Oracle(**kw)
    ctx = capi.Context(capi.param_make(**kw))
    S = raw.shape[0]
    d_r = dev(torch, raw, np.float32)
    d_bad = torch.zeros(S, n_beams, dtype=torch.uint8, device="cuda"); d_seg = torch.zeros(S, n_beams, dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    ctx.preprocess_batch(S, d_r, d_bad, d_seg); ctx.sync()
    pre = [o.preprocess(raw[k]) for k in range(S)]
    r = np.stack([q[0] for q in pre]); bad = np.stack([q[1] for q in pre]); seg = np.stack([q[2] for q in pre])
    assert same_bits(d_r.cpu().numpy(), r.astype(np.float32))
    assert np.array_equal(d_bad.cpu().numpy(), bad.astype(np.uint8)) and np.array_equal(d_seg.cpu().numpy(), seg.astype(np.int16))
    ref_idx = np.array([0, 1, 2, 3, 4, 5, 6, 7, 8, 0, 2, 5]); act_idx = np.array([1, 2, 3, 4, 5, 6, 7, 8, 9, 3, 0, 9])
    for which in (0, 1):
        out = run_match(torch, ctx, which, r, bad, seg, ref_idx, act_idx, want_err=False)
        pose_o, st_o = o.match_batch(which, r, bad, seg, ref_idx, act_idx)
        assert np.array_equal(out["status"], st_o)
        assert same_bits(out["pose"], pose_o)
    ctx.close()


@pytest.mark.parametrize("name", ["mit", "intel", "syn1081"])
def test_fmatch_chain_host_api(name, torch_mod):
    g = load(name)
    ctx, _ = make_ctx(name, g)
    n = g["raw_r"].shape[1]
    cur = [g["raw_r"][0].copy(), g["raw_bad"][0].copy(), np.zeros(n, np.int32)]
    B = min(20, g["fmatch_pose"].shape[0])
    for b in range(B):
        act = [g["raw_r"][b + 1].copy(), g["raw_bad"][b + 1].copy(), np.zeros(n, np.int32)]
        pose, pair, r2, a2 = ctx.fmatch_host(cur, act, b == 0)
        assert same_bits(pose, g["fmatch_pose"][b]), b
        assert same_bits(pair, g["fmatch_pair"][b]), (b, pair, g["fmatch_pair"][b])
        assert same_bits(a2[0], g["pre_r"][b + 1]) and np.array_equal(a2[2], g["pre_seg"][b + 1])
        cur = a2
    # single-pair host matcher + error index
    ref = (g["pre_r"][0], g["pre_bad"][0], g["pre_seg"][0]); act = (g["pre_r"][1], g["pre_bad"][1], g["pre_seg"][1])
    for which, nm in ((0, "psm"), (1, "icp")):
        st, pose, ret = ctx.match_host(which, ref, act)
        assert st == g[nm + "_status"][0] and same_bits(pose, g[nm + "_pose"][0])
        if st == 0:
            exy, na = ctx.error_index2_host(ref, act, (0, 0, 0), pose)
            assert na == g[nm + "_nassoc"][0] and same_bits(exy, g[nm + "_exy"][0])
    ctx.close()


def test_failure_status_codes(torch_mod):
    """Empty / featureless scans must fail like the reference's `throw 1` paths."""
    torch = torch_mod
    from laser_slam_amd import capi
    p = capi.param_preset("LMS511")
    ctx = capi.Context(p)
    n = p.points
    r = np.full((2, n), p.max_range + 1, np.float32)          # nothing in range
    d_r = dev(torch, r, np.float32); d_bad = torch.zeros(2, n, dtype=torch.uint8, device="cuda")
    d_seg = torch.zeros(2, n, dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    ctx.preprocess_batch(2, d_r, d_bad, d_seg)
    ctx.sync()
    out_p = run_match(torch, ctx, 0, d_r.cpu().numpy(), d_bad.cpu().numpy(), d_seg.cpu().numpy(), [0], [1])
    out_i = run_match(torch, ctx, 1, d_r.cpu().numpy(), d_bad.cpu().numpy(), d_seg.cpu().numpy(), [0], [1])
    assert out_p["status"][0] == capi.ST_NO_ORIENT
    assert out_i["status"][0] == capi.ST_FEW_POINTS
    assert np.all(out_p["pose"] == 0) and np.all(out_i["pose"] == 0)
    ctx.close()


@pytest.mark.parametrize("name", ["mit", "fr079"])
def test_grid_vs_golden(name, torch_mod):
    torch = torch_mod
    from laser_slam_amd import capi
    z = np.load(os.path.join(GOLDEN, "pmap_%s.npz" % name))
    G = z["geom"]
    ctx = capi.Context(capi.param_preset("LMS511"))
    grid = capi.Grid(ctx, int(G[0]), int(G[1]), int(G[2]), int(G[3]), float(G[4]), float(G[5]), float(G[6]))
    S, nb = z["gx"].shape
    # batched device path
    grid.update_batch(S, nb, dev(torch, z["gx"], np.float32), dev(torch, z["gy"], np.float32),
                      dev(torch, z["robot"][:, :2], np.float32), dev(torch, z["robot"][:, 2], np.int32))
    grid.compute_prob()
    hit, sm, prob = grid.layers()
    assert np.array_equal(hit, z["hit"].astype(np.float32))
    assert np.array_equal(sm, z["sum"].astype(np.float32))
    assert same_bits(prob, z["prob"])
    cx, cy = grid.cloud(0.95)
    assert same_bits(cx, z["cloud_x"]) and same_bits(cy, z["cloud_y"])
    # host one-scan-at-a-time path (CPMap::updateMap signature)
    grid.clear()
    for k in range(S):
        nv = int(z["robot"][k, 2])
        grid.update_host(z["gx"][k, :nv], z["gy"][k, :nv], float(z["robot"][k, 0]), float(z["robot"][k, 1]))
    hit2, sm2, _ = grid.layers()
    assert np.array_equal(hit2, hit) and np.array_equal(sm2, sm)
    grid.close(); ctx.close()


def test_grid_vs_oracle_random(torch_mod):
    """Random rays incl. out-of-map endpoints, degenerate rays, negative coordinates."""
    torch = torch_mod
    from laser_slam_amd import capi
    from oracle_lib import OraclePMap
    rng = np.random.default_rng(3)
    sx, sy, cx, cy, offx, offy, res = 300, 200, 120, 70, 1.25, -2.5, 0.05
    ctx = capi.Context(capi.param_preset("LMS211"))
    grid = capi.Grid(ctx, sx, sy, cx, cy, offx, offy, res)
    o = OraclePMap(sx, sy, cx, cy, offx, offy, res)
    S, nb = 24, 181
    gx = np.zeros((S, nb), np.float32); gy = np.zeros((S, nb), np.float32); rob = np.zeros((S, 2), np.float32)
    for s in range(S):
        rob[s] = rng.uniform(-4, 8, 2)
        ang = rng.uniform(0, 2 * np.pi, nb); d = rng.uniform(0, 12, nb)
        gx[s] = rob[s, 0] + d * np.cos(ang); gy[s] = rob[s, 1] + d * np.sin(ang)
        if s == 3:
            gx[s] = rob[s, 0]; gy[s] = rob[s, 1]          # all rays degenerate
        o.update(gx[s], gy[s], float(rob[s, 0]), float(rob[s, 1]))
    grid.update_batch(S, nb, dev(torch, gx, np.float32), dev(torch, gy, np.float32), dev(torch, rob, np.float32))
    grid.compute_prob(); o.compute()
    for a, b in zip(grid.layers(), o.layers()):
        assert same_bits(a, b)
    for _ in range(200):
        px, py = [float(np.float32(v)) for v in rng.uniform(-9, 9, 2)]
        assert grid.cell(px, py) == o.cell(px, py)
    grid.close(); ctx.close()


def test_grid_batch_ragged_scans(torch_mod):
    """Scans with different numbers of valid beams (incl. none) through the batched path against the oracle."""
    torch = torch_mod
    from laser_slam_amd import capi
    from oracle_lib import OraclePMap
    rng = np.random.default_rng(13)
    sx, sy, cx, cy, res = 400, 300, 200, 150, 0.05
    ctx = capi.Context(capi.param_preset("LMS151"))
    grid = capi.Grid(ctx, sx, sy, cx, cy, 0.0, 0.0, res)
    o = OraclePMap(sx, sy, cx, cy, 0.0, 0.0, res)
    S, nb = 40, 541
    nv = rng.integers(0, nb + 1, S).astype(np.int32); nv[3] = 0; nv[5] = nb; nv[7] = 1
    rob = rng.uniform(-6, 6, (S, 2)).astype(np.float32)
    ang = rng.uniform(0, 2 * np.pi, (S, nb)); d = rng.uniform(0, 14, (S, nb))
    gx = (rob[:, :1] + d * np.cos(ang)).astype(np.float32); gy = (rob[:, 1:] + d * np.sin(ang)).astype(np.float32)
    for s in range(S):
        if nv[s]:
            o.update(gx[s, :nv[s]], gy[s, :nv[s]], float(rob[s, 0]), float(rob[s, 1]))
    torch.cuda.synchronize()
    grid.update_batch(S, nb, dev(torch, gx, np.float32), dev(torch, gy, np.float32), dev(torch, rob, np.float32),
                      dev(torch, nv, np.int32))
    grid.compute_prob(); o.compute()
    for a, b in zip(grid.layers(), o.layers()):
        assert same_bits(a, b)
    grid.close(); ctx.close()


def test_full_size_batch_properties(torch_mod):
    """BASELINE-size batch (65 536 pairs of 1081-beam scans): results must not depend on a pair's
    position in the batch or on what runs beside it (every copy of a pair gives the same bits),
    a sample agrees with the CPU oracle, and the status histogram is stable."""
    torch = torch_mod
    from laser_slam_amd import capi, synth
    from oracle_lib import SYN1081, Oracle
    raw = np.concatenate([synth.make_scans(64, noise_seed=12345 + j)[0] for j in range(2)], 0)
    S, n = raw.shape
    ctx = capi.Context(capi.param_make(**synth.SYN1081))
    d_r = dev(torch, raw, np.float32); d_bad = torch.zeros(S, n, dtype=torch.uint8, device="cuda")
    d_seg = torch.zeros(S, n, dtype=torch.int16, device="cuda")
    torch.cuda.synchronize()
    ctx.preprocess_batch(S, d_r, d_bad, d_seg); ctx.sync()
    base = np.array([(s * 64 + k - 1, s * 64 + k) for s in range(2) for k in range(1, 64)], np.int32)     # 126 pairs
    B = 65536
    rng = np.random.default_rng(9)
    sel = rng.integers(0, len(base), B)
    r, bad, seg = d_r.cpu().numpy(), d_bad.cpu().numpy(), d_seg.cpu().numpy().astype(np.uint16)
    o = Oracle(**SYN1081)
    for which in (0, 1):
        out = run_match(torch, ctx, which, r, bad, seg, base[sel, 0], base[sel, 1], want_err=False)
        first = {}
        for b in range(B):
            k = int(sel[b])
            if k not in first:
                first[k] = b
        ref_pose = np.stack([out["pose"][first[int(k)]] for k in sel])
        assert same_bits(out["pose"], ref_pose)
        assert np.array_equal(out["status"], np.array([out["status"][first[int(k)]] for k in sel]))
        ks = sorted(first)[:12]
        pose_o, st_o = o.match_batch(which, r, bad.astype(np.int32), seg.astype(np.int32), base[ks, 0], base[ks, 1])
        got = np.stack([out["pose"][first[k]] for k in ks])
        assert same_bits(got, pose_o) and np.array_equal(st_o, [out["status"][first[k]] for k in ks])
    ctx.close()


def test_full_size_grid_checksum(torch_mod):
    """4096^2 grid, 2048 scans of 1081 beams: every in-range beam contributes exactly two hits
    (sum(hit) = 2 * valid beams) and one visit per traversed cell; inserting the
    scans in two halves or all at once gives identical layers."""
    torch = torch_mod
    from laser_slam_amd import capi, synth
    raw, _ = synth.make_scans(256)
    p = capi.param_make(**synth.SYN1081)
    fi = capi.param_tables(p)[0]
    nb = 2048
    poses = synth.path_pose(np.arange(nb))
    rr = np.concatenate([raw] * 8)[:nb] / 100.0
    gx = (poses[:, 0:1] + rr * np.cos(poses[:, 2:3] + fi[None])).astype(np.float32)
    gy = (poses[:, 1:2] + rr * np.sin(poses[:, 2:3] + fi[None])).astype(np.float32)
    ctx = capi.Context(p)
    g1 = capi.Grid(ctx, 4096, 4096, 1024, 1024, 0.0, 0.0, 0.05)
    g2 = capi.Grid(ctx, 4096, 4096, 1024, 1024, 0.0, 0.0, 0.05)
    dgx, dgy = dev(torch, gx, np.float32), dev(torch, gy, np.float32)
    drob = dev(torch, poses[:, :2], np.float32)
    torch.cuda.synchronize()
    g1.update_batch(nb, 1081, dgx, dgy, drob)
    h = nb // 2
    g2.update_batch(h, 1081, dgx[:h].contiguous(), dgy[:h].contiguous(), drob[:h].contiguous())
    g2.update_batch(nb - h, 1081, dgx[h:].contiguous(), dgy[h:].contiguous(), drob[h:].contiguous())
    hit1, sum1 = g1.download(0), g1.download(1)
    hit2, sum2 = g2.download(0), g2.download(1)
    assert np.array_equal(hit1, hit2) and np.array_equal(sum1, sum2)
    # the per-scan kernels (global atomics) and the owned-tile batch path (none) must agree,
    # whatever the number of workgroups per tile
    for env in ({"LSA_GRID_PATH": "scan"}, {"LSA_GRID_SPLITS": "1"}, {"LSA_GRID_SPLITS": "5"}):
        os.environ.update(env)
        try:
            g3 = capi.Grid(ctx, 4096, 4096, 1024, 1024, 0.0, 0.0, 0.05)
            g3.update_batch(nb, 1081, dgx, dgy, drob)
            assert np.array_equal(g3.download(0), hit1) and np.array_equal(g3.download(1), sum1), env
            g3.close()
        finally:
            for k in env:
                del os.environ[k]
    assert hit1.sum(dtype=np.float64) == 2.0 * nb * 1081          # the map is large enough for every ray
    # visits: one per traversed cell = max(|dx|,|dy|)+1 per ray (cells from the grid's own cell function)
    cx = np.array([[g1.cell(float(gx[s, i]), float(gy[s, i])) for i in range(0, 1081, 40)] for s in range(0, nb, 64)])
    assert sum1.sum(dtype=np.float64) > hit1.sum(dtype=np.float64)
    assert cx.min() >= 0 and cx.max() < 4096
    g1.close(); g2.close(); ctx.close()


def test_slam_loop_mit_4096(torch_mod):
    """BASELINE config 5 in small: the head of mit-cscail.log through the whole loop - FMatch's matcher (pm_icp) on
    consecutive scans, poses chained in double like OrientedPoint2D::oplus, every scan's end points put into a 4096^2
    counting grid at 5 cm at the chained pose - device against the same loop run with the CPU oracle.  Poses and
    relative motions must carry the same bits, hence the end points, hence the three grid layers."""
    torch = torch_mod
    from laser_slam_amd import capi
    from oracle_lib import Oracle, OraclePMap
    name = "mit"
    g = load(name)
    ctx, _ = make_ctx(name, g)
    laser = SETS[name][0]
    gp = golden_param(g)
    o = Oracle(laser); o.set_fimin_maxrange(np.float32(gp["fi_min"]), int(gp["max_range"]))
    r, bad, seg = g["pre_r"], g["pre_bad"], g["pre_seg"]
    S, n = r.shape
    ref_idx = np.arange(S - 1); act_idx = ref_idx + 1
    out = run_match(torch, ctx, 1, r, bad, seg, ref_idx, act_idx, want_err=False)
    rel_o = np.zeros((S - 1, 3), np.float32); st_o = np.zeros(S - 1, np.int32)
    for k in range(S - 1):
        st_o[k], rel_o[k], _, _ = o.match(1, (r[k], bad[k], seg[k]), (r[k + 1], bad[k + 1], seg[k + 1]))
    assert np.array_equal(out["status"], st_o) and same_bits(out["pose"][st_o == 0], rel_o[st_o == 0])

    def chain(rel, status):                     # global poses (m, m, rad) from relative ones (cm, cm, rad); failed pairs: no motion
        P = np.zeros((S, 3), np.float64)
        for k in range(S - 1):
            dx, dy, dth = (rel[k, 0] * 0.01, rel[k, 1] * 0.01, rel[k, 2]) if status[k] == 0 else (0.0, 0.0, 0.0)
            c, s_ = np.cos(P[k, 2]), np.sin(P[k, 2])
            P[k + 1] = [P[k, 0] + c * dx - s_ * dy, P[k, 1] + s_ * dx + c * dy, P[k, 2] + dth]
        return P

    def endpoints(P):
        fi = g["tab_fi"].astype(np.float64)
        ok = (bad == 0)
        rr = np.where(ok, r, 0.0).astype(np.float64) * 0.01
        gx = (P[:, 0:1] + rr * np.cos(P[:, 2:3] + fi[None])).astype(np.float32)
        gy = (P[:, 1:2] + rr * np.sin(P[:, 2:3] + fi[None])).astype(np.float32)
        return gx, gy, P[:, :2].astype(np.float32)

    gx_d, gy_d, rob_d = endpoints(chain(out["pose"], out["status"]))
    gx_o, gy_o, rob_o = endpoints(chain(rel_o, st_o))
    assert same_bits(gx_d, gx_o) and same_bits(gy_d, gy_o)
    geom = (4096, 4096, 2048, 2048, 0.0, 0.0, 0.05)
    grid = capi.Grid(ctx, *geom)
    torch.cuda.synchronize()
    grid.update_batch(S, n, dev(torch, gx_d, np.float32), dev(torch, gy_d, np.float32), dev(torch, rob_d, np.float32))
    grid.compute_prob()
    om = OraclePMap(*geom)
    for k in range(S):
        om.update(gx_o[k], gy_o[k], float(rob_o[k, 0]), float(rob_o[k, 1]))
    om.compute()
    for a, b in zip(grid.layers(), om.layers()):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert grid.layers()[1].sum() > 20 * S * n          # the rays really are tens of cells long
    grid.close(); ctx.close()


def test_grid_survives_wild_end_points(torch_mod):
    """End points that are not finite or astronomically far away have no defined cell in the reference (its cast of
    the cell number overflows); the device clamps cell numbers to +-2^28 so that the walks' integer arithmetic stays
    defined: the update must complete, sane scans beside the wild ones keep their counts, and nothing lands outside."""
    torch = torch_mod
    from laser_slam_amd import capi
    from oracle_lib import OraclePMap
    rng = np.random.default_rng(77)
    geom = (300, 260, 150, 130, 0.0, 0.0, 0.05)
    S, nb = 32, 200
    rob = rng.uniform(-3, 3, (S, 2)).astype(np.float32)
    ang = rng.uniform(0, 2 * np.pi, (S, nb)); d = rng.uniform(0.1, 6, (S, nb))
    gx = (rob[:, :1] + d * np.cos(ang)).astype(np.float32); gy = (rob[:, 1:] + d * np.sin(ang)).astype(np.float32)
    sane = OraclePMap(*geom)
    for s in range(S):
        sane.update(gx[s], gy[s], float(rob[s, 0]), float(rob[s, 1]))
    wx, wy = gx.copy(), gy.copy()
    wild = np.array([np.nan, np.inf, -np.inf, 1e30, -1e30, 3e9, -3e9, 1e12], np.float32)
    wx[::2, :8] = wild; wy[1::2, :8] = wild[::-1]; wx[5, 8:16] = wild; wy[5, 8:16] = wild
    ctx = capi.Context(capi.param_preset("LMS151"))
    grid = capi.Grid(ctx, *geom)
    torch.cuda.synchronize()
    grid.update_batch(S, nb, dev(torch, wx, np.float32), dev(torch, wy, np.float32), dev(torch, rob, np.float32))
    ctx.sync()
    hit, summ = grid.layers()[0], grid.layers()[1]
    assert np.isfinite(hit).all() and np.isfinite(summ).all()
    # every cell has at least the counts of the rays that were left alone
    keep = np.ones((S, nb), bool); keep[::2, :8] = False; keep[1::2, :8] = False; keep[5, 8:16] = False
    part = OraclePMap(*geom)
    for s in range(S):
        part.update(gx[s][keep[s]], gy[s][keep[s]], float(rob[s, 0]), float(rob[s, 1]))
    assert (summ >= part.layers()[1]).all() and (hit >= part.layers()[0]).all()
    assert summ.sum() < sane.layers()[1].sum() + S * 16 * (300 + 260)          # a wild ray adds a line across the map at most
    # the per-scan kernels (fewer than 16 scans) enter such a line where it reaches the map instead of walking 2^28 cells
    import time
    g2 = capi.Grid(ctx, *geom)
    t0 = time.time()
    g2.update_batch(4, nb, dev(torch, wx[:4], np.float32), dev(torch, wy[:4], np.float32), dev(torch, rob[:4], np.float32))
    ctx.sync()
    assert time.time() - t0 < 5.0
    part4 = OraclePMap(*geom)
    for s in range(4):
        part4.update(gx[s][keep[s]], gy[s][keep[s]], float(rob[s, 0]), float(rob[s, 1]))
    assert (g2.layers()[1] >= part4.layers()[1]).all() and (g2.layers()[0] >= part4.layers()[0]).all()
    g2.close(); grid.close(); ctx.close()


def test_contexts_are_reentrant_across_threads(torch_mod):
    """one matcher object per thread, several objects running concurrently.  Four threads, each with its
    own context (own HIP stream and scratch), match, preprocess and score at the same time; every thread must get
    the bits a lone context gets."""
    import threading
    torch = torch_mod
    from laser_slam_amd import capi, synth
    raw = np.concatenate([synth.make_scans(40, noise_seed=12345 + j)[0] for j in range(2)], 0)
    S, n = raw.shape
    base = np.array([(s * 40 + k - 1, s * 40 + k) for s in range(2) for k in range(1, 40)], np.int32)
    B = 2048
    sel = np.arange(B) % len(base)

    def work(which, out, key):
        ctx = capi.Context(capi.param_make(**synth.SYN1081))
        d_r = torch.from_numpy(raw).cuda(); d_bad = torch.zeros(S, n, dtype=torch.uint8, device="cuda"); d_seg = torch.zeros(S, n, dtype=torch.int16, device="cuda")
        torch.cuda.synchronize()
        ctx.preprocess_batch(S, d_r, d_bad, d_seg)
        ri = torch.from_numpy(np.ascontiguousarray(base[sel, 0])).cuda(); ai = torch.from_numpy(np.ascontiguousarray(base[sel, 1])).cuda()
        pose = torch.zeros(B, 3, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        for _ in range(3):
            ctx.match_batch(which, B, d_r, d_bad, d_seg, ri, ai, pose, d_status=st)
            ctx.sync()
        out[key] = (pose.cpu().numpy().copy(), st.cpu().numpy().copy())
        ctx.close()

    ref = {}
    work(1, ref, "icp"); work(0, ref, "psm")
    res = {}
    threads = [threading.Thread(target=work, args=(w, res, (w, t))) for t in range(2) for w in (0, 1)]
    for t in threads: t.start()
    for t in threads: t.join()
    assert len(res) == 4
    for (w, t), (pose, st) in res.items():
        want = ref["icp" if w == 1 else "psm"]
        assert np.array_equal(st, want[1]) and same_bits(pose, want[0]), (w, t)


@pytest.mark.parametrize("beams,window", [(1081, 200), (361, 40), (181, 20)])
def test_orientation_search_block_edges(beams, window, torch_mod):
    """The bounded orientation search evaluates the shifts in blocks of 64 around zero shift and abandons blocks
    whose partial sums exceed the zero block's minimum.  Rotations that put the minimum on a block edge, inside an
    outer block, at the ends of the window, a scan without valid readings and sparse scans must all give the
    oracle's status and bits (pm_orientation_search, src/zhpsm/ZHPolar_Match.cpp:1152-1261)."""
    from laser_slam_amd import capi, synth
    from oracle_lib import Oracle
    kw = dict(points=beams, fov_deg=270, max_range=3000, min_valid=beams // 4, window=window, fi_min_deg=-135, fi_max_deg=135)
    raw = synth.make_scans(6, noise_seed=synth.WORLD_SEED + 77, **({} if beams == 1081 else dict(n_beams=beams)))[0]
    o = Oracle(**kw)
    ctx = capi.Context(capi.param_make(**kw))
    rng = np.random.default_rng(5)
    W = window
    shifts = sorted({s for s in (-W, -W + 1, -161, -160, -97, -96, -95, -34, -33, -32, -31, -1, 0, 1, 30, 31, 32, 33, 95, 96, 97,
                                 159, 160, W - 1, W) if -W <= s <= W})
    checked = 0
    for k in range(3):
        ref_r, ref_bad, _ = o.preprocess(raw[k])
        for s in shifts:
            for sparse in (False, True):
                new_r = np.roll(ref_r, -s).astype(np.float32) + rng.normal(0, 0.5, beams).astype(np.float32)
                new_bad = np.roll(ref_bad, -s).astype(np.int32)
                if s > 0: new_bad[beams - s:] = 1          # readings rolled in from the other end
                if s < 0: new_bad[:-s] = 1
                if sparse:
                    new_bad[rng.random(beams) < 0.9] = 1
                want = o.orientation_search(ref_r, ref_bad, new_r, new_bad)
                got = ctx.orientation_search_host(ref_r, ref_bad, new_r, new_bad)
                assert (got[0] != 0) == (want[0] != 0), (beams, s, sparse)
                if want[0] == 0:
                    assert np.float32(got[1]).tobytes() == np.float32(want[1]).tobytes(), (beams, s, sparse, got, want)
                checked += 1
    # ranges beyond 2^19 cm: the search leaves its clamp form (terms would no longer stay below 1 after scaling)
    ref_r, ref_bad, _ = o.preprocess(raw[1])
    big_ref = (ref_r * np.float32(1000.0)).astype(np.float32)
    big_new = (np.roll(big_ref, -3) + rng.normal(0, 500.0, beams)).astype(np.float32)
    assert big_ref[ref_bad == 0].max() > 2.0 ** 19
    want = o.orientation_search(big_ref, ref_bad, big_new, np.roll(ref_bad, -3).astype(np.int32))
    got = ctx.orientation_search_host(big_ref, ref_bad, big_new, np.roll(ref_bad, -3).astype(np.int32))
    assert (got[0] != 0) == (want[0] != 0)
    if want[0] == 0:
        assert np.float32(got[1]).tobytes() == np.float32(want[1]).tobytes()
    # nothing valid on one side: no orientation
    ref_r, ref_bad, _ = o.preprocess(raw[0])
    allbad = np.ones(beams, np.int32)
    assert o.orientation_search(ref_r, ref_bad, ref_r, allbad)[0] != 0
    assert ctx.orientation_search_host(ref_r, ref_bad, ref_r, allbad)[0] != 0
    assert ctx.orientation_search_host(ref_r, allbad, ref_r, ref_bad)[0] != 0
    ctx.close()
    assert checked == 6 * len(shifts)


def test_contexts_on_concurrent_host_threads(torch_mod):
    """One matcher object per thread is the reference's threading model (threadLocal1.cpp:20,144): four host threads,
    each with its own context (stream, pair counter, scratch), match the same batch at the same time with both matchers;
    every thread gets the single-threaded result bit for bit."""
    import threading
    from laser_slam_amd import capi, synth
    torch = torch_mod
    raw = synth.make_scans(40, noise_seed=synth.WORLD_SEED + 9)[0]
    p = capi.param_make(**synth.SYN1081)
    S, n = raw.shape
    B = 4096
    idx = (np.arange(B) % (S - 1)).astype(np.int32)

    def run(out, slot):
        ctx = capi.Context(p)
        d_r = torch.from_numpy(raw).cuda(); d_bad = torch.zeros(S, n, dtype=torch.uint8, device="cuda")
        d_seg = torch.zeros(S, n, dtype=torch.int16, device="cuda")
        ri = torch.from_numpy(idx).cuda(); ai = ri + 1
        torch.cuda.synchronize()
        ctx.preprocess_batch(S, d_r, d_bad, d_seg)
        res = []
        for which in (capi.MATCH_ICP, capi.MATCH_PSM):
            pose = torch.zeros(B, 3, device="cuda"); st = torch.zeros(B, dtype=torch.int32, device="cuda")
            torch.cuda.synchronize()
            for _ in range(3):
                ctx.match_batch(which, B, d_r, d_bad, d_seg, ri, ai, pose, d_status=st)
            ctx.sync()
            res.append((pose.cpu().numpy().copy(), st.cpu().numpy().copy()))
        ctx.close()
        out[slot] = res

    ref = [None]
    run(ref, 0)
    outs = [None] * 4
    threads = [threading.Thread(target=run, args=(outs, k)) for k in range(4)]
    for t in threads: t.start()
    for t in threads: t.join()
    for res in outs:
        assert res is not None
        for (pose, st), (pose0, st0) in zip(res, ref[0]):
            assert np.array_equal(st, st0) and np.array_equal(pose.view(np.uint32), pose0.view(np.uint32))
