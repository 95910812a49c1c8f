#!/usr/bin/env python3
"""Instruction mix of kernels in a gfx950 assembly listing (hipcc -S --cuda-device-only):  tools/isa_mix.py vpgpu.s k_sumfold3b_multi [k_leaf_hash ...]
Counts static instructions per kernel by class; loops are not weighted (read the hot loop's labels in the listing for that)."""
import re, sys, collections
src = open(sys.argv[1]).read().split("\n")
want = sys.argv[2:]
cur = None; mix = {}
for ln in src:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        name = m.group(1); cur = None
        for w in want:
            if w in name: cur = name; mix[cur] = collections.Counter()
        continue
    if cur is None: continue
    if ln.startswith("\t.") or ln.startswith(".L") and ln.endswith(":"): 
        if ".end_amdhsa_kernel" in ln or "s_endpgm" in ln: pass
        continue
    t = ln.strip().split()
    if not t or t[0].startswith(";") or t[0].startswith("."): continue
    op = t[0]
    if not re.match(r"^(v_|s_|ds_|global_|buffer_|flat_|scratch_)", op): continue
    mix[cur][op] += 1
for k, c in mix.items():
    tot = sum(c.values())
    valu = sum(v for o, v in c.items() if o.startswith("v_"))
    mad = sum(v for o, v in c.items() if o.startswith("v_mad_u64") or o.startswith("v_mul_"))
    print("%s\n  total %d  VALU %d  multiplier %d  SALU %d  LDS %d  vmem %d" % (k[:100], tot, valu, mad, sum(v for o, v in c.items() if o.startswith("s_")),
          sum(v for o, v in c.items() if o.startswith("ds_")), sum(v for o, v in c.items() if re.match(r"^(global_|buffer_|flat_|scratch_)", o))))
    print("  " + "  ".join("%s %d" % (o, v) for o, v in c.most_common(28)))
