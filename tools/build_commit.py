#!/usr/bin/env python3
"""Diagnostic: build liblsa_core of a given commit into build_ab/<name>.so (for bisecting with the soak tools)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
for arg in sys.argv[1:]:
    name, _, commit = arg.partition("=")
    d = "/tmp/commit_" + name
    subprocess.check_call("rm -rf %s && mkdir -p %s/laser_slam_amd/csrc %s/include" % (d, d, d), shell=True)
    files = subprocess.check_output(["git", "ls-tree", "-r", "--name-only", commit, "laser_slam_amd/csrc", "include"], cwd=ROOT, text=True).split()
    for f in files:
        open(os.path.join(d, f), "wb").write(subprocess.check_output(["git", "show", "%s:%s" % (commit, f)], cwd=ROOT))
    os.makedirs(os.path.join(ROOT, "build_ab"), exist_ok=True)
    subprocess.check_call(["hipcc"] + ge.HIP_FLAGS + ["-o", os.path.join(ROOT, "build_ab", name + ".so"),
                           os.path.join(d, "laser_slam_amd/csrc/core.hip"), os.path.join(d, "laser_slam_amd/csrc/carmen.cpp")],
                          stderr=subprocess.DEVNULL)
    print(name, commit, "built")
