#!/usr/bin/env python3
"""Loop structure and static instruction mix of a slice of an llvm-objdump listing (e.g. the sweeps of the step kernel,
between its two s_setprio).  usage: isa_loops.py slice.txt <hex address of the kernel symbol>"""
import re, sys, collections
L = open(sys.argv[1]).read().split('\n')
base = int(sys.argv[2], 16) if len(sys.argv) > 2 else 0          # address of the kernel symbol (branch targets are symbol + offset)
addr = {}
for i, l in enumerate(L):
    m = re.search(r'// ([0-9A-F]{12}):', l)
    if m: addr[int(m.group(1), 16)] = i
br = []
for i, l in enumerate(L):
    m = re.search(r'\b(s_cbranch_\w+|s_branch)\s.*\+0x([0-9a-f]+)>', l)
    if m and base + int(m.group(2), 16) in addr:
        br.append((i, m.group(1), addr[base + int(m.group(2), 16)]))
labels = addr
def cls(op):
    if op.startswith("v_"):
        if "f64" in op: return "f64"
        if "_f32" in op or "pk_" in op: return "f32"
        if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")): return "lane"
        if op.startswith("v_cndmask"): return "cnd"
        if op.startswith("v_mov"): return "mov"
        if op.startswith("v_cmp"): return "cmp"
        return "vint"
    if op.startswith("s_"):
        if op.startswith(("s_waitcnt", "s_nop")): return "wait"
        if op.startswith(("s_cbranch", "s_branch")): return "branch"
        return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith(("global_", "flat_", "buffer_")): return "mem"
    return "other"
def count(a, b):
    c = collections.Counter()
    for l in L[a:b]:
        t = l.split()
        if not t or t[0].endswith(':') or re.match(r'^[0-9a-f]+$', t[0]): continue
        c[cls(t[0])] += 1
    return c
print("lines", len(L), "labels", len(labels))
print("whole slice", dict(count(0, len(L))))
for i, _, t in br:
    if t < i:
        print("loop %d..%d (%d lines): %s" % (t, i, i - t, dict(count(t, i))))
