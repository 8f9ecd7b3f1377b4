#!/bin/bash
O=gpurun_out/r3l; mkdir -p $O
export ARBSTEP_LIB=build/ab/${V:-r3pk2}.so
timeout -k 10 300 python tools/pack_fused_check.py 2>&1 | grep -v amdgpu | tee $O/check.txt
B="python bench.py --no-cpu-baseline --no-per-step-leg --min-seconds 2"
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms'])"; }
for r in 1 2; do
  for b in 4096 8192 16384 65536; do
    echo -n "batch $b auto(no pack): "; ARB_FORCE_PACK=0 timeout -k 10 120 $B --batch $b 2>/dev/null | val
    echo -n "batch $b pack: "; ARB_FORCE_PACK=1 timeout -k 10 120 $B --batch $b 2>/dev/null | val
  done
  echo -n "config5 no pack: "; ARB_FORCE_PACK=0 timeout -k 10 120 $B --config 5 2>/dev/null | val
  echo -n "config5 pack: "; ARB_FORCE_PACK=1 timeout -k 10 120 $B --config 5 2>/dev/null | val
done | tee $O/bench.txt
