#!/usr/bin/env python3
"""Where does a kernel spill?  python tools/isa_spills.py <file.s> <mangled kernel name substring>
Lists, per basic block of the kernel's ISA (hipcc --cuda-device-only -S), the scratch loads / stores and SGPR<->VGPR-lane moves, the
backward branches (loops) and the instruction count, so that spills inside a loop can be told from spills around it."""
import re
import sys

src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and key in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
labels = {}
blocks = []   # (label, first_line, n_instr, n_sst, n_sld, n_wl, n_rl)
cur = ["entry", 0, 0, 0, 0, 0, 0]
order = []
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(tuple(cur))
        cur = [m.group(1), i, 0, 0, 0, 0, 0]
        labels[m.group(1)] = len(blocks)
        continue
    s = l.strip()
    if not s or s.startswith(";") or s.startswith("."):
        continue
    cur[2] += 1
    if s.startswith("scratch_store"):
        cur[3] += 1
    elif s.startswith("scratch_load"):
        cur[4] += 1
    elif s.startswith("v_writelane"):
        cur[5] += 1
    elif s.startswith("v_readlane"):
        cur[6] += 1
blocks.append(tuple(cur))
idx = {b[0]: k for k, b in enumerate(blocks)}
# backward branches
loops = []
for k, b in enumerate(blocks):
    first = b[1]
    nxt = blocks[k + 1][1] if k + 1 < len(blocks) else len(body)
    for l in body[first:nxt]:
        m = re.search(r"s_cbranch_\w+\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
        if m:
            tgt = m.group(1) or m.group(2)
            if tgt in idx and idx[tgt] <= k:
                loops.append((idx[tgt], k))
tot = [sum(b[c] for b in blocks) for c in range(2, 7)]
print("kernel %s: %d blocks, %d instructions, scratch_store %d, scratch_load %d, v_writelane %d, v_readlane %d" % (key, len(blocks), *tot))
# innermost-loop membership depth per block
depth = [0] * len(blocks)
for a, b in loops:
    for k in range(a, b + 1):
        depth[k] += 1
print("loops (block ranges, instruction count, scratch st/ld inside):")
for a, b in sorted(set(loops), key=lambda x: (x[0], -x[1])):
    n = sum(blocks[k][2] for k in range(a, b + 1))
    st = sum(blocks[k][3] for k in range(a, b + 1))
    ld = sum(blocks[k][4] for k in range(a, b + 1))
    wl = sum(blocks[k][5] for k in range(a, b + 1))
    rl = sum(blocks[k][6] for k in range(a, b + 1))
    if n >= 200 or st or ld:
        print("  blocks %4d-%4d  instr %6d  scratch st %4d ld %4d  writelane %4d readlane %4d" % (a, b, n, st, ld, wl, rl))
