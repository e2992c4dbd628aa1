#!/usr/bin/env python3
"""Diagnostic (GPU box): bench.py's single-call latency entry alone (CPolarMatch::FMatch over consecutive log scans), for the
product's library or a variant: python3 tools/fmatch_lat.py [build_ab/<variant>.so | -]   (LSA_ICP_WIDE=0/1 forces a width)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from laser_slam_amd import capi
if len(sys.argv) > 1 and sys.argv[1] != "-": capi.LIB_PATH = os.path.abspath(sys.argv[1])
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
raw = bench.build_scan_set()
res = bench.secondary_latency(torch, capi, raw)["fmatch_latency"]
for k, v in res.items():
    if not isinstance(v, dict) or "CPolarMatch_FMatch" not in v:
        print(k, v); continue
    print(sys.argv[1] if len(sys.argv) > 1 else "-", k, v["beams"], "CPolarMatch::FMatch median %.0f us p95 %.0f" % (v["CPolarMatch_FMatch"]["median_us"], v["CPolarMatch_FMatch"]["p95_us"]), flush=True)
