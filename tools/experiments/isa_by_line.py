#!/usr/bin/env python3
"""Static instruction counts per source line of one kernel in an `llvm-objdump -d -l` listing (build with
-gline-tables-only, e.g. tools/spill_map.sh): VALU float32 / float64 / other, SALU, LDS, scratch -- to see what the
loop body of the sweeps is made of.  usage: tools/isa_by_line.py dis.txt <kernel-substring> [file-substring [lo [hi]]]"""
import re, sys, collections
dis, kern = sys.argv[1], sys.argv[2]
fsub = sys.argv[3] if len(sys.argv) > 3 else ""
lo = int(sys.argv[4]) if len(sys.argv) > 4 else 0
hi = int(sys.argv[5]) if len(sys.argv) > 5 else 10 ** 9
cnt = collections.defaultdict(lambda: collections.Counter())
inside, cur = False, ("?", 0)
for ln in open(dis):
    if re.match(r"^[0-9a-f]+ <", ln):
        inside = kern in ln
        continue
    if not inside:
        continue
    m = re.match(r"^; (.*):(\d+)$", ln.strip())
    if m:
        cur = (m.group(1).split("/")[-1], int(m.group(2)))
        continue
    t = ln.split()
    if not t or t[0].startswith(";") or ":" in t[0] and len(t) == 1:
        continue
    op = t[0]
    if op.startswith("v_"):
        if "f64" in op: c = "f64"
        elif "_f32" in op or "pk_" in op: c = "f32"
        elif op.startswith("v_mfma"): c = "mfma"
        elif op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")): c = "lane"
        elif op.startswith("v_cndmask"): c = "cnd"
        elif op.startswith("v_mov") or op.startswith("v_accvgpr"): c = "mov"
        elif op.startswith("v_cmp"): c = "cmp"
        else: c = "vint"
        if "dpp" in ln or "quad_perm" in ln or "row_" in ln: c += "+dpp"
    elif op.startswith("s_"):
        c = "salu" if not op.startswith(("s_waitcnt", "s_nop", "s_cbranch", "s_branch")) else ("wait" if op.startswith(("s_waitcnt", "s_nop")) else "branch")
    elif op.startswith("ds_"): c = "lds"
    elif op.startswith("scratch_"): c = "scratch"
    elif op.startswith(("global_", "flat_", "buffer_")): c = "mem"
    else: c = "other"
    cnt[cur][c] += 1
tot = collections.Counter()
rows = []
for (f, l), c in cnt.items():
    if fsub in f and lo <= l <= hi:
        rows.append((f, l, c)); tot.update(c)
rows.sort()
keys = sorted(tot, key=lambda k: -tot[k])
print("total", dict(tot))
for f, l, c in rows:
    if sum(c.values()) >= (int(sys.argv[6]) if len(sys.argv) > 6 else 12):
        print("%s:%d  %d  %s" % (f, l, sum(c.values()), " ".join("%s=%d" % (k, c[k]) for k in keys if c[k])))
