import sys, os, ctypes as C, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
torch.zeros(1, device="cuda")
from helpers import SETS, golden_param, load
import oracle_lib
from oracle_lib import plicp
from laser_slam_amd import capi
capi.LIB_PATH = os.path.abspath("build_ab/trace.so")
name, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
laser, fn, tag = SETS[name]
g = load(name); gp = golden_param(g)
p = capi.param_preset(laser); p.fi_min = float(np.float32(gp["fi_min"])); p.fi_max = float(np.float32(gp["fi_max"])); p.max_range = int(gp["max_range"])
raw = np.ascontiguousarray(g["raw_r"], np.float32)
tb = np.zeros(8 * 128); 
oracle_lib._orc().orc_plicp_set_trace(tb.ctypes.data_as(C.c_void_p), 128)
w = plicp(raw[a], raw[b], p.fi_min, p.dfi, p.min_range, p.max_range)
no = oracle_lib._orc().orc_plicp_trace_count()
ctx = capi.Context(p)
got = capi.plicp_host(ctx, raw[a], raw[b])
td = np.zeros(8 * 128); nd = C.c_int(0)
capi.lib().lsa_debug_plicp_trace(td.ctypes.data_as(C.c_void_p), C.byref(nd))
np.set_printoptions(linewidth=250, precision=12)
print("oracle", w); print("device", got)
for k in range(max(no, nd.value)):
    o_ = tb[8*k:8*k+8]; d_ = td[8*k:8*k+8]
    print(k, "O", o_); print(k, "D", d_)
    if not np.array_equal(o_[4:], d_[4:]) : 
        print("   ^^^ first discrete difference"); break
