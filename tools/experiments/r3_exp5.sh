#!/bin/bash
# A/B of quick builds against the built library at 4096 and 65536 worlds (three-wave builds picked by batch size), + bitwise
O=gpurun_out/r3i; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-per-step-leg --min-seconds 2"
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms'])"; }
ARBSTEP_LIB=build/ab/r3base.so python tools/xlib_dump.py $O/base.npz quick > /dev/null 2>&1
for v in $VARIANTS; do ARBSTEP_LIB=build/ab/$v.so python tools/xlib_dump.py $O/$v.npz quick > /dev/null 2>&1; echo "== $v vs r3base: $(python tools/xlib_cmp.py $O/base.npz $O/$v.npz | tail -1)"; done
for r in 1 2; do
  echo -n "lib: "; timeout -k 10 120 $B 2>/dev/null | val
  for v in $VARIANTS; do echo -n "$v: "; ARBSTEP_LIB=build/ab/$v.so timeout -k 10 120 $B 2>/dev/null | val; done
  echo -n "lib 65536: "; timeout -k 10 120 $B --batch 65536 2>/dev/null | val
  for v in $VARIANTS; do echo -n "$v 65536: "; ARBSTEP_LIB=build/ab/$v.so timeout -k 10 120 $B --batch 65536 2>/dev/null | val; done
done 2>&1 | tee $O/bench.txt
