#!/usr/bin/env python3
"""Static instruction mix of ONE kernel per segment of the step (round 5: what the 9.0 k "other" vector instructions are).

Build the compiler's assembly output with comment markers at the phase boundaries,

    hipcc -S --cuda-device-only --offload-arch=gfx950 -O3 -ffp-contract=on -std=c++17 -I include \
          -DARB_QUICK=3 -DARB_DEVELOPMENT -DARB_MARKS -o build/tmp/marks.s arboris_python_amd/csrc/arb_kernels.hip

then:  tools/isa_phase_mix.py build/tmp/marks.s [kernel-substring] [--top N]

Markers (arb_kernels.hip, ARB_MARKS): P0 phase A .. A6/P1 phase A' .. P2 phase B (B3 subtree sums, B4 dof products, B5 rows
of Z, B6 constraint rows, B7 controllers) .. P3 phase C (C4 pivots, C5 +gvel) .. P4 phase D .. P5 block inverses, sweep
setup, then the two copies of the local solve (GS_gt0 decisions, gt1 alpha / block / c1, kappa, gt2 root, gt3 4x4 solve,
gt4 hand-over + velocity update) .. P6 phase E .. P7 store.  Straight-line code is executed once per step; the loops are
weighted by hand (level loops: maxdepth + 1 passes; the sweeps: 80 solves).  Classes: f32, f64 (arithmetic), cvt, mov
(v_mov / accvgpr), cnd (v_cndmask), cmp, lane (v_readlane / readfirstlane / writelane), dpp (any DPP-modified vector op),
vint (integer / logic / shifts), mfma, salu, branch, wait (s_waitcnt / s_nop), lds, scratch, mem.
"""
import collections
import re
import sys

path = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "IfLi44ELi1ELi0ELi4ELi2E"
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 0


def classify(op, line):
    if op.startswith("v_"):
        if "dpp" in line or "quad_perm" in line or "row_shr" in line or "row_bcast" in line or "wave_shr" in line:
            return "dpp"
        if op.startswith("v_mfma") or op.startswith("v_smfma"):
            return "mfma"
        if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
            return "lane"
        if op.startswith("v_cndmask"):
            return "cnd"
        if op.startswith(("v_mov", "v_accvgpr", "v_swap")):
            return "mov"
        if op.startswith("v_cmp"):
            return "cmp"
        if op.startswith("v_cvt"):
            return "cvt"
        if "f64" in op:
            return "f64"
        if "f32" in op or op.startswith("v_pk_"):
            return "f32"
        return "vint"
    if op.startswith("s_"):
        if op.startswith(("s_waitcnt", "s_nop", "s_sleep")):
            return "wait"
        if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm")):
            return "branch"
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("global_", "flat_", "buffer_")):
        return "mem"
    return "other"


seg = collections.OrderedDict()
ops = collections.defaultdict(collections.Counter)
inside = False
cur = None
nseg = collections.Counter()
for ln in open(path):
    if re.match(r"^[A-Za-z_][\w$.]*:\s*(;.*)?$", ln) and not ln.startswith((".L", "\t")):
        inside = kern in ln
        cur = "prologue"
        continue
    if not inside:
        continue
    t = ln.strip()
    if t.startswith("; ARB_MARK"):
        name = t.split()[2]
        nseg[name] += 1
        cur = name if nseg[name] == 1 else "%s#%d" % (name, nseg[name])
        continue
    if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
        continue
    if t.startswith("s_endpgm"):
        inside = False
    op = t.split()[0]
    c = classify(op, t)
    seg.setdefault(cur, collections.Counter())[c] += 1
    ops[cur][op] += 1

cols = ["f32", "f64", "cvt", "mov", "cnd", "cmp", "lane", "dpp", "vint", "mfma", "salu", "branch", "wait", "lds", "scratch", "mem"]
print("%-12s %6s | %s" % ("segment", "valu", " ".join("%6s" % c for c in cols)))
tot = collections.Counter()
for name, c in seg.items():
    valu = sum(c[k] for k in ("f32", "f64", "cvt", "mov", "cnd", "cmp", "lane", "dpp", "vint", "mfma"))
    print("%-12s %6d | %s" % (name, valu, " ".join("%6d" % c[k] for k in cols)))
    tot.update(c)
    if top:
        print("             top:", ", ".join("%s %d" % kv for kv in ops[name].most_common(top)))
valu = sum(tot[k] for k in ("f32", "f64", "cvt", "mov", "cnd", "cmp", "lane", "dpp", "vint", "mfma"))
print("%-12s %6d | %s" % ("TOTAL", valu, " ".join("%6d" % tot[k] for k in cols)))
