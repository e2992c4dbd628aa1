"""Synthetic 1081-beam scan generator (SURVEY.md §8d "synthetic world").

20 m x 15 m room with 12 axis-aligned boxes, robot on a Lissajous path,
analytic ray casting, Gaussian range noise (sigma 1 cm), 1 mm quantisation,
30 m clip, 270 deg field of view.  Ranges are returned in centimetres in the
reader format of the reference (CARMEN ingest, ZHPolar_Match.cpp:250-259):
raw `r`, `bad = 0`, and `r < 10 cm -> max_range + 1`.

This is input plumbing (numpy only); it carries no matcher arithmetic.
"""
import numpy as np

ROOM_W, ROOM_H = 20.0, 15.0
N_BOXES = 12
WORLD_SEED = 12345

# parameter block injected for the 1081-beam sensor (no preset in the reference)
SYN1081 = dict(points=1081, fov_deg=270, max_range=3000, min_valid=300, window=200,
               fi_min_deg=-135, fi_max_deg=135)


def _world():
    """Returns wall segments as an (M,4) array of x0,y0,x1,y1 [m]."""
    rng = np.random.Generator(np.random.MT19937(WORLD_SEED))
    segs = [(0, 0, ROOM_W, 0), (ROOM_W, 0, ROOM_W, ROOM_H), (ROOM_W, ROOM_H, 0, ROOM_H), (0, ROOM_H, 0, 0)]
    for _ in range(N_BOXES):
        w = rng.uniform(0.3, 1.5)
        h = rng.uniform(0.3, 1.5)
        cx = rng.uniform(1.5, ROOM_W - 1.5)
        cy = rng.uniform(1.5, ROOM_H - 1.5)
        x0, x1, y0, y1 = cx - w / 2, cx + w / 2, cy - h / 2, cy + h / 2
        segs += [(x0, y0, x1, y0), (x1, y0, x1, y1), (x1, y1, x0, y1), (x0, y1, x0, y0)]
    return np.asarray(segs, np.float64)


def path_pose(k):
    """Robot pose (x, y, theta) [m, rad] of scan k on the Lissajous path."""
    t = 0.02 * np.asarray(k, np.float64)
    return np.stack([10 + 6 * np.sin(t), 7.5 + 4 * np.sin(2 * t + 0.3), 0.5 * t], -1)


def _raycast(segs, ox, oy, ang):
    """Analytic ray/segment intersection; ox,oy scalars, ang (N,). -> ranges (N,) [m]."""
    dx, dy = np.cos(ang)[:, None], np.sin(ang)[:, None]
    x0, y0, x1, y1 = (segs[None, :, i] for i in range(4))
    ex, ey = x1 - x0, y1 - y0
    den = dx * ey - dy * ex
    with np.errstate(divide="ignore", invalid="ignore"):
        t = ((x0 - ox) * ey - (y0 - oy) * ex) / den      # along the ray
        u = ((x0 - ox) * dy - (y0 - oy) * dx) / den      # along the segment
    ok = (np.abs(den) > 1e-12) & (t > 1e-9) & (u >= 0) & (u <= 1)
    t = np.where(ok, t, np.inf)
    return t.min(axis=1)


def make_scans(n_scans, n_beams=1081, fov_deg=270.0, noise_seed=WORLD_SEED, start=0,
               sigma_m=0.01, max_range_cm=3000):
    """Returns (r[S,N] float32 cm in reader format, poses[S,3] float64 m/rad)."""
    segs = _world()
    rng = np.random.Generator(np.random.MT19937(noise_seed))
    fov = np.deg2rad(fov_deg)
    bearings = -fov / 2 + fov * np.arange(n_beams) / (n_beams - 1)
    poses = path_pose(np.arange(start, start + n_scans))
    out = np.empty((n_scans, n_beams), np.float32)
    for s in range(n_scans):
        x, y, th = poses[s]
        d = _raycast(segs, x, y, th + bearings)
        d = d + rng.normal(0.0, sigma_m, n_beams)
        d = np.round(d * 1000.0) / 1000.0              # 1 mm quantisation
        d = np.clip(d, 0.0, 30.0)
        r = (d * 100.0).astype(np.float32)             # m -> cm
        r[r < 10.0] = np.float32(max_range_cm + 1)
        out[s] = r
    return out, poses
