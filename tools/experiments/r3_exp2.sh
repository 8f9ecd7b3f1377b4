#!/bin/bash
# round 3: LDS-aliased kernel (13.1 KB per wave) at 2 and 3 waves per SIMD against the round's baseline build; bitwise check
O=gpurun_out/r3c; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-per-step-leg --min-seconds 2"
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms'])"; }
for v in ${VARIANTS:-r3base r3lds r3lds3}; do
  ARBSTEP_LIB=build/ab/$v.so timeout -k 10 200 python tools/xlib_dump.py $O/dump_$v.npz quick > $O/dump_$v.log 2>&1 || tail -3 $O/dump_$v.log
done
for v in ${VARIANTS:-r3base r3lds r3lds3}; do echo "== $v vs r3base"; python tools/xlib_cmp.py $O/dump_r3base.npz $O/dump_$v.npz | tail -4; done
for r in 1 2; do
  for v in ${VARIANTS:-r3base r3lds r3lds3}; do echo -n "$v: "; ARBSTEP_LIB=build/ab/$v.so timeout -k 10 120 $B 2>/dev/null | val; done
  for v in ${VARIANTS:-r3base r3lds r3lds3}; do echo -n "$v 65536: "; ARBSTEP_LIB=build/ab/$v.so timeout -k 10 120 $B --batch 65536 2>/dev/null | val; done
done 2>&1 | tee $O/bench.txt
