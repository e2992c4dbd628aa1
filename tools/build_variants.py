#!/usr/bin/env python3
"""Diagnostic: build liblsa_core variants (one per argument `name=-DA,-DB`) into build_ab/<name>.so, in parallel."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
os.makedirs(os.path.join(ROOT, "build_ab"), exist_ok=True)
procs = []
for a in sys.argv[1:]:
    name, _, fl = a.partition("=")
    flags = [f for f in fl.split(",") if f]
    out = os.path.join(ROOT, "build_ab", name + ".so")
    procs.append((name, subprocess.Popen(["hipcc"] + ge.HIP_FLAGS + flags + ["-o", out, os.path.join(ge.CSRC, "core.hip"), os.path.join(ge.CSRC, "carmen.cpp")],
                                         stderr=open(os.path.join(ROOT, "build_ab", name + ".log"), "w"))))
for name, p in procs:
    print(name, "rc", p.wait())
