#!/usr/bin/env python3
"""Diagnostic (GPU box): bench.py's global-localisation entry alone."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from laser_slam_amd import capi, synth
raw = bench.build_scan_set()
p = capi.param_make(**synth.SYN1081); ctx = capi.Context(p, 0)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); ctx.set_stream(stream.cuda_stream)
print(json.dumps(bench.secondary_global(torch, capi, ctx, raw, p), indent=1))
