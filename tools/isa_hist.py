#!/usr/bin/env python3
"""Instruction-class histogram per kernel of a `hipcc -S --cuda-device-only` listing.
usage: tools/isa_hist.py file.s   (static counts; loops are counted once)"""
import re, sys, collections
kern = None
hist = collections.OrderedDict()
for line in open(sys.argv[1]):
    m = re.match(r"^(_Z\w+):", line)
    if m:
        kern = m.group(1); hist[kern] = collections.Counter(); continue
    if kern is None: continue
    if line.startswith(".Lfunc_end"):
        kern = None; continue
    m = re.match(r"^\s+([a-z_0-9]+)\s", line)
    if not m: continue
    op = m.group(1)
    if op.startswith("v_pk_"): cls = "v_pk"
    elif re.match(r"v_(fma|fmac|mul|add|sub|mac)_f32", op): cls = "v_f32"
    elif op.startswith("v_mov") or op.startswith("v_accvgpr"): cls = "v_mov"
    elif op.startswith("v_"): cls = "v_other"
    elif op.startswith("ds_"): cls = "ds"
    elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"): cls = "vmem"
    elif op.startswith("scratch_"): cls = "scratch"
    elif op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop"): cls = "s_wait"
    elif op.startswith("s_"): cls = "salu"
    else: cls = "other"
    hist[kern][cls] += 1
    hist[kern]["op:" + op] += 1
for k, h in hist.items():
    short = re.sub(r"^_ZN5smhip9sm_kernelINS_", "", k)[:60]
    tot = sum(v for c, v in h.items() if not c.startswith("op:"))
    print(f"{short:60s} total {tot:6d} " + " ".join(f"{c}={v}" for c, v in h.items() if not c.startswith("op:")))
    if len(sys.argv) > 2:
        top = sorted(((v, c[3:]) for c, v in h.items() if c.startswith("op:")), reverse=True)[:int(sys.argv[2])]
        print("      " + " ".join(f"{c}:{v}" for v, c in top))
