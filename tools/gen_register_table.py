#!/usr/bin/env python3
"""The register table of DESIGN.md section 3, generated from the SHIPPED library's code-object metadata so that it cannot go
stale (round 4's table said 332 / 337 / 405 spilled registers where the library had 399 / 402 / 401).

usage: tools/gen_register_table.py            print the table
       tools/gen_register_table.py --write    rewrite the block between the REGISTERS markers of DESIGN.md
       tools/gen_register_table.py --check    exit 1 when DESIGN.md's block differs from the library's numbers
`__graft_entry__.build()` runs --write after it has rebuilt the library; tests/test_capi_cpu.py runs --check.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "arboris_python_amd", "libarbstep.so")
DESIGN = os.path.join(ROOT, "DESIGN.md")
BEGIN, END = "<!-- REGISTERS:BEGIN (tools/gen_register_table.py) -->", "<!-- REGISTERS:END -->"

# (precision, column sets, mode, feat, cm) of the float32 / float64 44-row kernels worth a row, with what they are
ROWS = [
    ("f", 1, 0, 52, 2, "body-space columns compiled for four contacts, three waves -- the headline kernel (round 6)"),
    ("f", 1, 0, 53, 2, "... with user torques / torque sequences / running cost (the MPC leg at large batches)"),
    ("f", 1, 0, 52, 0, "body-space columns, four contacts, two waves"),
    ("f", 1, 0, 53, 0, "... with user torques (the 2048-rollout MPC leg)"),
    ("f", 1, 0, 4, 2, "classical columns specialised for four contacts, three waves (`classic_columns_f32`; the headline until round 5)"),
    ("f", 1, 0, 4, 0, "classical columns, four contacts, two waves"),
    ("f", 1, 0, 20, 2, "body-space columns, any number of contacts (eight: `contacts8`), three waves"),
    ("f", 1, 0, 20, 0, "body-space columns, two waves"),
    ("f", 1, 0, 19, 2, "body-space columns, every optional input (rollout logs), three waves"),
    ("f", 1, 0, 8, 2, "no constraints (config 2 at large batches), three waves"),
    ("f", 1, 0, 8, 0, "no constraints, two waves (config 2)"),
    ("f", 1, 0, 0, 2, "general, three waves (`general_kernel`)"),
    ("f", 1, 0, 1, 2, "general + user torques, three waves"),
    ("f", 1, 0, 3, 2, "general, every optional input, three waves"),
    ("f", 1, 0, 0, 0, "general, two waves"),
    ("f", 1, 0, 0, 3, "mixed build (float32 state, float64 elimination; ARB_STEP_MIXED)"),
    ("f", 2, 0, 0, 0, "general, two column sets"),
    ("f", 1, 1, 3, 0, "inspect"),
    ("f", 1, 1, 19, 0, "inspect, body-space columns"),
    ("d", 1, 0, 4, 0, "float64 specialised (`strict_f64`)"),
    ("d", 1, 0, 20, 0, "float64 body-space columns"),
    ("d", 1, 0, 0, 0, "float64 general"),
]


def kernel_stats(lib=LIB):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "so_stats_all.py"), lib], capture_output=True, text=True).stdout
    stats = {}
    for line in out.splitlines():
        m = re.match(r"arb_step_kernel<(\w)Li(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E\s+vgpr\s+(\d+) agpr\s+(\d+) \(spill\s+(\d+)\)\s+sgpr\s+(\d+) \(spill\s+(\d+)\)\s+scratch\s+(\d+) B", line)
        if m:
            key = (m.group(1), int(m.group(2)), int(m.group(3)), int(m.group(4)), int(m.group(5)), int(m.group(6)))
            stats[key] = tuple(int(x) for x in m.groups()[6:])
    return stats


def table():
    st = kernel_stats()
    if not st:
        raise SystemExit("no kernel metadata found in %s (is the library built? are the llvm tools in /opt/rocm/lib/llvm/bin?)" % LIB)
    lines = ["| kernel `arb_step_kernel<T, 44, NSETS, MODE, FEAT, CM>` | VGPR | spilled VGPR | spilled SGPR | scratch per lane |", "|---|---|---|---|---|"]
    for p, ns, mode, feat, cm, what in ROWS:
        k = (p, 44, ns, mode, feat, cm)
        if k not in st:
            continue
        vg, ag, vs, sg, ss, scr = st[k]
        lines.append("| %s `<%s, 44, %d, %d, %d, %d>` | %d | %d | %d | %d B |" % (what, "float" if p == "f" else "double", ns, mode, feat, cm, vg, vs, ss, scr))
    lines.append("")
    lines.append("(%d `arb_step_kernel` instantiations in the shipped library; spilled SGPRs go to VGPR lanes, not to memory.)" % len(st))
    return "\n".join(lines)


def current_block():
    s = open(DESIGN).read()
    if BEGIN not in s or END not in s:
        return None, s
    i0 = s.index(BEGIN) + len(BEGIN)
    return s[i0:s.index(END)].strip("\n"), s


if __name__ == "__main__":
    t = table()
    if "--write" in sys.argv:
        cur, s = current_block()
        if cur is None:
            raise SystemExit("DESIGN.md has no REGISTERS markers")
        i0 = s.index(BEGIN) + len(BEGIN)
        open(DESIGN, "w").write(s[:i0] + "\n" + t + "\n" + s[s.index(END):])
        print("DESIGN.md register table rewritten")
    elif "--check" in sys.argv:
        cur, _ = current_block()
        if cur != t:
            print("DESIGN.md's register table is stale: run tools/gen_register_table.py --write")
            sys.exit(1)
        print("register table up to date")
    else:
        print(t)
