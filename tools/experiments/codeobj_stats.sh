#!/bin/bash
# Register / spill metadata and static instruction mix of the device kernels inside a host object.
# usage: tools/codeobj_stats.sh build/obj/part_float_44.o [kernel-name-substring]
set -e
B=/opt/rocm/lib/llvm/bin
OBJ=$1; PAT=${2:-}
T=$(mktemp -d)
$B/llvm-objcopy -O binary --only-section=.hip_fatbin "$OBJ" $T/fb
$B/clang-offload-bundler --unbundle --type=o --input=$T/fb --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/co
$B/llvm-readelf --notes $T/co | awk '
  /\.name:/ {name=$2}
  /\.sgpr_count:/ {sg=$2} /\.sgpr_spill_count:/ {ss=$2} /\.vgpr_count:/ {vg=$2} /\.vgpr_spill_count:/ {vs=$2}
  /\.group_segment_fixed_size:/ {lds=$2} /\.private_segment_fixed_size:/ {pr=$2}
  /\.wavefront_size:/ {printf "%s\n    vgpr %s (spill %s)  sgpr %s (spill %s)  scratch %s B  static lds %s B\n", name, vg, vs, sg, ss, pr, lds}' | grep -A1 -- "$PAT" | grep -v "^--"
if [ -n "$PAT" ]; then
  $B/llvm-objdump -d --no-show-raw-insn $T/co > $T/dis
  for k in $(grep -E "^[0-9a-f]+ <.*>:" $T/dis | sed 's/.*<\(.*\)>:/\1/' | grep -- "$PAT"); do
    echo "== $k"
    awk -v k="<$k>:" '$2==k {on=1; next} /^[0-9a-f]+ <.*>:/ {on=0} on && NF>0 {print $1}' $T/dis | sort | uniq -c | sort -rn > $T/mix
    echo "   total $(awk '{s+=$1} END {print s}' $T/mix)  mfma $(grep -c mfma $T/mix || true)"
    awk '{n=$1; i=$2; if (i ~ /^v_.*f64/) f64+=n; else if (i ~ /^v_pk_/) pk+=n; else if (i ~ /^v_.*f32/) f32+=n;
          if (i ~ /^v_readlane|^v_readfirstlane/) rl+=n; if (i ~ /^v_writelane/) wl+=n; if (i=="s_nop") nop+=n;
          if (i ~ /^ds_/) ds+=n; if (i ~ /^s_load|^s_buffer_load/) sl+=n; if (i ~ /^scratch_/) sc+=n; if (i ~ /mfma/) mf+=n;
          if (i ~ /^s_waitcnt/) wc+=n}
         END {printf "   f64 %d  f32 %d  pk_f32 %d  readlane %d  writelane %d  s_nop %d  ds %d  s_load %d  scratch %d  mfma %d  s_waitcnt %d\n", f64,f32,pk,rl,wl,nop,ds,sl,sc,mf,wc}' $T/mix
  done
fi
rm -rf $T
