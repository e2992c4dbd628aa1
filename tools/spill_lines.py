#!/usr/bin/env python3
"""Build container: where a kernel's scratch stores / loads come from.  usage: spill_lines.py <asm built with
-gline-tables-only> <mangled kernel prefix>: per source line (innermost frame) the number of scratch stores and loads behind
the first loop header, with the loop depth the assembler's comments give for the enclosing block."""
import collections, re, sys
lines = open(sys.argv[1]).read().split("\n")
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
for s in [i for i, l in enumerate(lines) if l.startswith(sys.argv[2]) and ":" in l.split(";")[0]]:
    e = next(i for i in range(s, len(lines)) if lines[i].startswith(".Lfunc_end"))
    cur, depth, seen_loop = None, 0, False
    st, ld = collections.Counter(), collections.Counter()
    for l in lines[s:e]:
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
        m = re.search(r"Loop Header: Depth=(\d+)|in Loop: Header=\S+ Depth=(\d+)", l)
        if m:
            depth = int(m.group(1) or m.group(2)); seen_loop = True
        if seen_loop and "scratch_store" in l: st[(cur, depth)] += 1
        if seen_loop and "scratch_load" in l: ld[(cur, depth)] += 1
    print(lines[s].split(":")[0][:70], "in-loop stores", sum(st.values()), "loads", sum(ld.values()))
    for k, v in sorted(st.items(), key=lambda x: -x[1])[:20]: print("   store x%d  %s:%d  depth %d" % (v, k[0][0], k[0][1], k[1]))
    for k, v in sorted(ld.items(), key=lambda x: -x[1])[:12]: print("   load  x%d  %s:%d  depth %d" % (v, k[0][0], k[0][1], k[1]))
