#!/usr/bin/env python3
"""Histogram of scratch (spill) instructions per 20-line source region for the plain float32 44-row step kernel of a
build made by tools/spill_map.sh: usage tools/spill_hist.py build/spill_<name>/dis.txt [substring of the mangled kernel
name: Li44ELi1ELi0ELi0ELi0E = two-wave build (default), ...Li2E = three waves]"""
import re, sys, collections
on = False; cur = None; hist = collections.Counter(); n = ns = 0
for l in open(sys.argv[1]):
    m = re.match(r'^[0-9a-f]+ <(.*)>:', l)
    if m:
        on = (sys.argv[2] if len(sys.argv) > 2 else 'Li44ELi1ELi0ELi0ELi0E') in m.group(1); continue
    if not on: continue
    m = re.match(r'^; .*/(arb_\w+\.(?:hip|h)):(\d+)', l)
    if m: cur = (m.group(1), int(m.group(2))); continue
    if l.startswith(';') or not l.strip(): continue
    n += 1
    if 'scratch_' in l:
        ns += 1; hist[(cur[0], cur[1] // 20 * 20, 'L' if 'scratch_load' in l else 'S')] += 1
print("scratch ops", ns, "of", n, "instructions")
for key in sorted(hist): print("  %s:%d-%d %s x%d" % (key[0], key[1], key[1] + 19, key[2], hist[key]))
