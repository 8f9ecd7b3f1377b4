#!/usr/bin/env python3
"""Register / spill / LDS metadata of EVERY device kernel of a shared library whose .hip_fatbin holds several offload
bundles (libarbstep.so: one per translation unit).  usage: tools/so_stats_all.py lib.so [name-substring]"""
import os, re, subprocess, sys, tempfile
B = "/opt/rocm/lib/llvm/bin/"
lib, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "arb_step_kernel")
with tempfile.TemporaryDirectory() as td:
    fb = os.path.join(td, "fb")
    subprocess.check_call([B + "llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fb])
    data = open(fb, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), data)]
    for k, st in enumerate(starts):
        blob = data[st:(starts[k + 1] if k + 1 < len(starts) else len(data))]
        bf, co = os.path.join(td, "b%d" % k), os.path.join(td, "co%d" % k)
        open(bf, "wb").write(blob)
        if subprocess.call([B + "clang-offload-bundler", "--unbundle", "--type=o", "--input=" + bf,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co],
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) != 0 or not os.path.exists(co):
            continue
        notes = subprocess.run([B + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        cur = {}
        for line in notes.splitlines():
            m = re.match(r"\s*-?\s*\.(\w+):\s*(.*)", line.strip())
            if not m:
                continue
            key, val = m.group(1), m.group(2).strip()
            if key == "name" and val.startswith("_Z"):
                cur = {"name": val}
            elif key in ("sgpr_count", "sgpr_spill_count", "vgpr_count", "vgpr_spill_count", "agpr_count", "private_segment_fixed_size"):
                cur[key] = val
            elif key == "wavefront_size" and cur.get("name") and pat in cur["name"]:
                short = re.sub(r"EvPK8DevModel.*", "", cur["name"]).replace("_Z15arb_step_kernelI", "arb_step_kernel<")
                print("%-44s vgpr %3s agpr %3s (spill %3s)  sgpr %3s (spill %3s)  scratch %4s B" % (
                    short, cur.get("vgpr_count"), cur.get("agpr_count", "0"), cur.get("vgpr_spill_count"),
                    cur.get("sgpr_count"), cur.get("sgpr_spill_count"), cur.get("private_segment_fixed_size")))
