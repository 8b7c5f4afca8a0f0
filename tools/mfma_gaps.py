#!/usr/bin/env python3
"""Reads hipcc -S output, finds the basic blocks with the most MFMAs in one kernel and prints, for each, how the other
instructions are spread between consecutive MFMAs (a matrix instruction leaves two vector-issue slots free; longer
runs of vector work between two MFMAs stall the matrix pipe).  usage: mfma_gaps.py file.s kernel-name-substring"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l.split(":")[0] and ":" in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blocks, cur = [], []
for l in lines[start:end]:
    t = l.strip()
    if not t or t.startswith(";"):
        continue
    if re.match(r"^\.?[A-Za-z_0-9$]+:", t):
        blocks.append(cur)
        cur = []
        continue
    op = t.split()[0]
    cur.append(op)
    if op.startswith("s_cbranch") or op == "s_branch":
        blocks.append(cur)
        cur = []
blocks.append(cur)
blocks = [b for b in blocks if sum(o.startswith("v_mfma") for o in b) >= 10]
blocks.sort(key=lambda b: -sum(o.startswith("v_mfma") for o in b))
for b in blocks[:3]:
    n_mfma = sum(o.startswith("v_mfma") for o in b)
    gaps, g = [], collections.Counter()
    for o in b:
        if o.startswith("v_mfma"):
            gaps.append(g)
            g = collections.Counter()
        else:
            kind = "valu" if o.startswith("v_") else "lds" if o.startswith("ds_") else "vmem" if o.startswith(("global_", "buffer_")) else "wait" if o.startswith("s_waitcnt") else "nop" if o == "s_nop" else "salu"
            g[kind] += 1
    valu = [x["valu"] for x in gaps[1:]]
    hist = collections.Counter(valu)
    print(f"block: {len(b)} instructions, {n_mfma} MFMA, valu {sum(valu)}, lds {sum(x['lds'] for x in gaps)}, vmem {sum(x['vmem'] for x in gaps)}, "
          f"waitcnt {sum(x['wait'] for x in gaps)}, s_nop {sum(x['nop'] for x in gaps)}")
    print("  valu per MFMA gap: " + ", ".join(f"{k}:{hist[k]}" for k in sorted(hist)))
    extra = sum(max(0, v - 2) for v in valu)
    print(f"  vector instructions beyond two per gap: {extra}")
