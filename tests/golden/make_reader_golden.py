#!/usr/bin/env python3
"""Fixtures for the two text readers besides CARMEN: synthetic input files in the formats of
CPolarMatch::readRawSeed (ZHPolar_Match.cpp:123-170: "timestamp,n,offset,r_1,..,r_n", metres) and
construct511Data (:649-667: "timestamp n r_1 .. r_n", metres), and what the REFERENCE's own readers - compiled in
place into oracle/_ref - make of them.  Writes readers_rawseed.csv, readers_511.txt, readers.npz next to itself.
Runs only where /root/reference exists; the fixtures are data."""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from oracle_lib import Ref
import ctypes as C

rng = np.random.default_rng(2024)

def ranges(n):
    r = rng.uniform(0.3, 45.0, n)
    r[rng.integers(0, n, 6)] = rng.choice([0.0, 0.05, 0.1, 0.1000001, 0.0999, 81.0], 6)      # at / below the 10 cm minimum, beyond the maximum
    return r

# RawSeed: LMS211 (181 beams); one record with the wrong count is skipped, at most 20 lines are looked at
n = 181
lines = []
for k in range(8):
    r = ranges(n)
    if k == 3:
        lines.append("1222.5,%d,0,%s" % (n - 1, ",".join("%.3f" % v for v in r[:-1])))
    lines.append("%.6f,%d,0,%s" % (1222.0 + 0.1 * k, n, ",".join(("%.3f" if k % 2 else "%.6f") % v for v in r)))
open(os.path.join(HERE, "readers_rawseed.csv"), "w").write("\n".join(lines) + "\n")
ref = Ref("LMS211")
S = ref.L.ref_read_rawseed(C.c_void_p(ref.h), os.path.join(HERE, "readers_rawseed.csv").encode())
assert S == 8
f32, i32 = np.float32, np.int32
out = {}
rr = np.zeros((S, n), f32); xx = np.zeros((S, n), f32); yy = np.zeros((S, n), f32); bb = np.zeros((S, n), i32); pose = np.zeros((S, 3), f32)
for k in range(S):
    ref.L.ref_get_scan(C.c_void_p(ref.h), k, rr[k].ctypes.data_as(C.c_void_p), xx[k].ctypes.data_as(C.c_void_p),
                       yy[k].ctypes.data_as(C.c_void_p), bb[k].ctypes.data_as(C.c_void_p), pose[k].ctypes.data_as(C.c_void_p))
out.update(rawseed_r=rr, rawseed_x=xx, rawseed_y=yy, rawseed_bad=bb)

# 511 text: LMS511 (361 beams), one record per line; a line with another count is refused
n5 = 361
ref5 = Ref("LMS511")
lines5, r5, x5, y5, b5, ok5 = [], [], [], [], [], []
for k in range(6):
    r = ranges(n5)
    cnt = n5 if k != 2 else n5 - 2
    line = "%d %d %s" % (1000 + k, cnt, " ".join("%.4f" % v for v in r[:cnt]))
    lines5.append(line)
    a, b, c, d = np.zeros(n5, f32), np.zeros(n5, f32), np.zeros(n5, f32), np.zeros(n5, i32)
    ok = ref5.L.ref_construct511(C.c_void_p(ref5.h), line.encode(), a.ctypes.data_as(C.c_void_p), b.ctypes.data_as(C.c_void_p),
                                 c.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p))
    ok5.append(ok); r5.append(a); x5.append(b); y5.append(c); b5.append(d)
open(os.path.join(HERE, "readers_511.txt"), "w").write("\n".join(lines5) + "\n")
out.update(t511_ok=np.array(ok5, i32), t511_r=np.stack(r5), t511_x=np.stack(x5), t511_y=np.stack(y5), t511_bad=np.stack(b5))
# SICK text: LMS511 (361 beams), "tag n r_1 .. r_n"; every record is complete (a shorter one would keep the previous
# record's tail - and, in the first record, uninitialised memory - in the reference)
refs = Ref("LMS511")
lines_s = []
for k in range(5):
    r = ranges(n5)
    lines_s.append("SICK %d %s" % (n5, " ".join("%.3f" % v for v in r)))
open(os.path.join(HERE, "readers_sick.txt"), "w").write("\n".join(lines_s) + "\n")
Ss = refs.L.ref_read_sick(C.c_void_p(refs.h), os.path.join(HERE, "readers_sick.txt").encode())
assert Ss == 5
rs = np.zeros((Ss, n5), f32); xs = np.zeros((Ss, n5), f32); ys = np.zeros((Ss, n5), f32); bs = np.zeros((Ss, n5), i32); ps = np.zeros((Ss, 3), f32)
for k in range(Ss):
    refs.L.ref_get_scan(C.c_void_p(refs.h), k, rs[k].ctypes.data_as(C.c_void_p), xs[k].ctypes.data_as(C.c_void_p),
                        ys[k].ctypes.data_as(C.c_void_p), bs[k].ctypes.data_as(C.c_void_p), ps[k].ctypes.data_as(C.c_void_p))
out.update(sick_r=rs, sick_x=xs, sick_y=ys, sick_bad=bs)
# pm_is_corridor on the preprocessed scans of the golden sets (and on a few degenerate ones: it throws 3 / 2)
from helpers import SETS, golden_param, load
for name in ("mit", "fr079", "intel", "indoor"):
    laser = SETS[name][0]
    g = load(name); gp = golden_param(g)
    rc = Ref(laser); rc.set_fimin_maxrange(np.float32(gp["fi_min"]), int(gp["max_range"]))
    res = []
    for k in range(g["pre_r"].shape[0]):
        r_ = np.ascontiguousarray(g["pre_r"][k], f32); b_ = np.ascontiguousarray(g["pre_bad"][k], i32); s_ = np.ascontiguousarray(g["pre_seg"][k], i32)
        res.append(rc.L.ref_is_corridor(C.c_void_p(rc.h), r_.ctypes.data_as(C.c_void_p), b_.ctypes.data_as(C.c_void_p), s_.ctypes.data_as(C.c_void_p)))
    # a scan without any segment, and one with a single two-beam segment
    z = np.zeros(g["pre_r"].shape[1], i32); one = z.copy(); one[10:12] = 1
    r_ = np.ascontiguousarray(g["pre_r"][0], f32)
    for s_ in (z, one):
        res.append(rc.L.ref_is_corridor(C.c_void_p(rc.h), r_.ctypes.data_as(C.c_void_p), z.ctypes.data_as(C.c_void_p), s_.ctypes.data_as(C.c_void_p)))
    out["corridor_" + name] = np.array(res, i32)
    print(name, "corridor flags:", np.bincount(np.array(res) % 100 if False else np.clip(res, 0, 103), minlength=2)[:2], "throws:", [v for v in res if v >= 100])
# synthetic corridors and rooms (LMS511, 0..180 degrees): walls at +-w/2 beside the robot's heading, ends far away
rc = Ref("LMS511")
nn = 361
th = np.deg2rad(np.linspace(0, 180, nn))
syn_r, syn_b, syn_s, syn_res = [], [], [], []
for k, (w, yaw, noise) in enumerate([(200, 0.0, 0.0), (200, 0.3, 0.5), (160, -0.2, 1.0), (300, 0.1, 3.0), (800, 0.0, 1.0), (200, 1.2, 0.5)]):
    a = th - np.pi / 2 + yaw                                   # bearing relative to the corridor's axis
    r = np.minimum((w / 2) / np.maximum(np.abs(np.sin(a)), 1e-3), 1500.0 if k != 4 else 500.0)
    r = (r + rng.normal(0, noise, nn)).astype(f32)
    r2, b2, s2 = rc.preprocess(r)
    syn_r.append(r2); syn_b.append(b2); syn_s.append(s2)
    syn_res.append(rc.L.ref_is_corridor(C.c_void_p(rc.h), r2.ctypes.data_as(C.c_void_p), b2.ctypes.data_as(C.c_void_p), s2.ctypes.data_as(C.c_void_p)))
out.update(corridor_syn_r=np.stack(syn_r), corridor_syn_bad=np.stack(syn_b), corridor_syn_seg=np.stack(syn_s), corridor_syn=np.array(syn_res, i32))
print("synthetic corridor flags:", syn_res)
np.savez_compressed(os.path.join(HERE, "readers.npz"), **out)
print("rawseed scans", S, "| 511 lines parsed", ok5)
