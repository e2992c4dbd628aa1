#!/usr/bin/env python3
"""Diagnostic (build container, no GPU): static VALU instruction counts of one kernel, attributed to the
kernel-body source line each instruction was inlined into (the outermost frame of the .loc chain).
usage: hipcc ... -gline-tables-only --cuda-device-only -S -o core_g.s csrc/core.hip
       isa_lines.py core_g.s <mangled kernel prefix> <body file> [inner]"""
import re, sys, collections
asm, prefix, body = sys.argv[1], sys.argv[2], sys.argv[3]
inner = len(sys.argv) > 4
cnt = collections.Counter(); cnt_in = collections.Counter()
inside = False; cur = None; cur_in = None
for ln in open(asm):
    if ln.startswith(prefix): inside = True; continue
    if inside and ln.startswith(".Lfunc_end"): break
    if not inside: continue
    if re.match(r'\s*\.loc\s', ln):
        fr = re.findall(r'([A-Za-z0-9_\.]+):(\d+):\d+', ln)
        if fr:
            outer = [f for f in fr if f[0] == body]
            cur = int(outer[-1][1]) if outer else None
            cur_in = (fr[0][0], int(fr[0][1]))
        continue
    s = ln.strip()
    if s.startswith("v_"):
        cnt[cur] += 1; cnt_in[(cur, cur_in)] += 1
print("total VALU", sum(cnt.values()))
for k in sorted(cnt, key=lambda x: (x is None, x)): print("%5d  %s:%s" % (cnt[k], body, k))
if inner:
    for (k, v) in cnt_in.most_common(80): print("%5d  body %s <- %s:%s" % (v, k[0], k[1][0], k[1][1]))
