#!/usr/bin/env python3
"""Diagnostic (GPU box): bench.py's entries on the reference's own logs alone (all matchers, rates and iteration counts)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from laser_slam_amd import capi
if len(sys.argv) > 1 and sys.argv[1] != "-":
    capi.LIB_PATH = os.path.abspath(sys.argv[1])
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
res = bench.secondary_logs(torch, capi)
for k, e in res.items():
    if "beams" in e:
        print("%-12s %4d beams:" % (k, e["beams"]), "  ".join("%s %.3f M/s (%.1f it)" % (m, e[m]["value"] / 1e6, e[m].get("mean_iterations", 0)) for m in ("pm_icp", "pm_psm", "plicp", "zhicp_p2p") if m in e))
        print("             beam-iterations/s: pm_icp %.1f G, pm_psm %.1f G" % (e["pm_icp"]["beam_iterations_per_s"] / 1e9, e["pm_psm"]["beam_iterations_per_s"] / 1e9))
    else:
        print("%-12s %.3f M/s (%.1f it, %.0f points)" % (k, e["value"] / 1e6, e["mean_iterations"], e["points_per_problem"]))
