#!/bin/bash
# Register / spill metadata of the device kernels inside a built shared library (tools/quick_build.sh output or libarbstep.so).
# usage: tools/so_stats.sh build/ab/x.so [kernel-name-substring]
B=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$B/llvm-objcopy -O binary --only-section=.hip_fatbin "$1" $T/fb 2>/dev/null
$B/clang-offload-bundler --unbundle --type=o --input=$T/fb --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/co 2>/dev/null
$B/llvm-readelf --notes $T/co | awk '
  /\.name:/ {name=$2}
  /\.sgpr_count:/ {sg=$2} /\.sgpr_spill_count:/ {ss=$2} /\.vgpr_count:/ {vg=$2} /\.vgpr_spill_count:/ {vs=$2} /\.agpr_count:/ {ag=$2}
  /\.private_segment_fixed_size:/ {pr=$2}
  /\.wavefront_size:/ {printf "%s\n    vgpr %s agpr %s (spill %s)  sgpr %s (spill %s)  scratch %s B\n", name, vg, ag, vs, sg, ss, pr}' | grep -A1 -- "${2:-arb_step}" | grep -v "^--"
rm -rf $T
