#!/bin/bash
# -ffp-contract=on (source-level fused multiply-adds only) against the default build: determinism across builds and speed
O=gpurun_out/r3j; mkdir -p $O
sed -i 's/(torch.float32, torch.float64)/(torch.float32,)/' tools/gsw_pack_check.py
ARBSTEP_LIB=build/ab/r3on.so timeout -k 10 300 python tools/gsw_pack_check.py 2>&1 | grep -v amdgpu | tee $O/pack_check.txt
ARBSTEP_LIB=build/ab/r3on.so timeout -k 10 300 python tools/w23_diff.py 2>&1 | grep -v amdgpu | tee $O/w23.txt
B="python bench.py --no-cpu-baseline --no-per-step-leg --min-seconds 2"
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms'])"; }
for r in 1 2; do
  echo -n "lib: "; timeout -k 10 120 $B 2>/dev/null | val
  echo -n "contract=on: "; ARBSTEP_LIB=build/ab/r3on.so timeout -k 10 120 $B 2>/dev/null | val
  echo -n "lib W2: "; ARB_FORCE_WAVES=2 timeout -k 10 120 $B 2>/dev/null | val
  echo -n "contract=on W2: "; ARB_FORCE_WAVES=2 ARBSTEP_LIB=build/ab/r3on.so timeout -k 10 120 $B 2>/dev/null | val
done | tee $O/bench.txt
