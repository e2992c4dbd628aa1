#!/usr/bin/env python3
"""Build container, no GPU: static checks of how a kernel's assembly handles scalar registers that were spilled into
lanes of a vector register (v_writelane / v_readlane) - the mechanism behind round 2's pm_icp abort.
usage: sgpr_spill_check.py <asm file> <mangled kernel name prefix>
Reports, per function body found under the prefix: (1) lanes that are read back but never written; (2) scratch spills /
reloads of a lane-carrying VGPR that are not bracketed by `s_or_saveexec_b64 ..., -1` (a store under a partial EXEC
mask loses the inactive lanes' scalars); (3) ordinary vector instructions that write a lane-carrying VGPR; (4) read-backs
that precede every write of their lane in the layout order of a straight-line prologue."""
import re, sys
asm, prefix = sys.argv[1], sys.argv[2]
lines = open(asm).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith(prefix) and ":" in l.split(";")[0]]
if not start:
    sys.exit("kernel not found")
rc = 0
for s in start:
    e = next(i for i in range(s, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[s:e]
    wl = {}; rl = {}
    for i, l in enumerate(body):
        m = re.match(r"\s*v_writelane_b32 (v\d+), \S+, (\d+)", l)
        if m: wl.setdefault((m.group(1), int(m.group(2))), []).append(i)
        m = re.match(r"\s*v_readlane_b32 \S+, (v\d+), (\d+)", l)
        if m: rl.setdefault((m.group(1), int(m.group(2))), []).append(i)
    carriers = sorted({k[0] for k in wl})
    never = sorted(k for k in rl if k[0] in carriers and k not in wl)
    bad_scratch = []
    for i, l in enumerate(body):
        m = re.match(r"\s*scratch_(store|load)_dword\S* .*\b(v\d+)\b", l)
        if m and m.group(2) in carriers:
            prev = next((body[j] for j in range(i - 1, max(i - 4, 0), -1) if body[j].strip() and not body[j].strip().startswith(";")), "")
            if not re.search(r"s_or_saveexec_b64 \S+ -1", prev):
                bad_scratch.append((i, l.strip()))
    clobber = []
    for i, l in enumerate(body):
        m = re.match(r"\s*(v_\w+) (v\d+)\b", l)
        if m and m.group(2) in carriers and m.group(1) not in ("v_writelane_b32",) and not m.group(1).startswith("v_cmp"):
            clobber.append((i, l.strip()))
    print("%s: %d lines, lane carriers %s, %d lanes written, %d lanes read" % (lines[s].split(":")[0][:60], len(body), carriers, len(wl), len(rl)))
    print("  lanes read but never written:", never)
    print("  scratch traffic of a carrier outside whole-wave mode:", bad_scratch[:8])
    print("  vector writes to a carrier:", clobber[:8])
    if never or bad_scratch or clobber:
        rc = 1
sys.exit(rc)
