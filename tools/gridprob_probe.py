#!/usr/bin/env python3
"""Diagnostic (GPU box): the probability pass of a 4096^2 CPMap, cached (back to back) and cold (1 GiB written between launches)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from laser_slam_amd import capi, synth
if len(sys.argv) > 1: capi.LIB_PATH = os.path.abspath(sys.argv[1])
raw = bench.build_scan_set()
p = capi.param_make(**synth.SYN1081); ctx = capi.Context(p, 0)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); ctx.set_stream(stream.cuda_stream)
print({k: (round(v["value"] / 1e9, 1), round(v["roofline"]["achieved"])) for k, v in bench.secondary_grid(torch, capi, ctx, raw, p).items() if "prob" in k})
