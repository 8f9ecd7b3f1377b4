#!/bin/bash
# round 3, GPU experiments before the kernel work (run through gpurun from the repo root):
#   1. tools/valu_issue_probe: wave-instructions per cycle per SIMD at 1..4 resident waves
#   2. bench.py at 8 / 7 / 6 / 4 waves per CU (ARB_LDS_PAD pins the occupancy of the shipped kernel)
#   3. the sweeps cut to 10 (a timing-only build): upper bound of what packing two worlds per wavefront can return
#   4. tools/pack_model.py: packing efficiency and pair imbalance from the device's own decision traces
O=gpurun_out/r3b; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-per-step-leg --min-seconds 2"
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms'])"; }
timeout -k 10 120 ./build/valu_issue_probe > $O/valu_probe.txt 2>&1; echo "probe rc=$?"
for r in 1 2; do
  for pad in 0 1100 4100 14000; do echo -n "pad $pad: " ; ARB_LDS_PAD=$pad timeout -k 10 120 $B 2>/dev/null | val; done
done > $O/lds_pad.txt 2>&1
cat $O/lds_pad.txt
for r in 1 2; do
  for v in r3base r3gs10; do echo -n "$v: "; ARBSTEP_LIB=build/ab/$v.so timeout -k 10 120 $B 2>/dev/null | val; done
  for v in r3base r3gs10; do echo -n "$v 8 contacts: "; ARBSTEP_LIB=build/ab/$v.so timeout -k 10 120 $B --contacts 8 2>/dev/null | val; done
  for v in r3base r3gs10; do echo -n "$v 65536: "; ARBSTEP_LIB=build/ab/$v.so timeout -k 10 120 $B --batch 65536 2>/dev/null | val; done
done > $O/gs10.txt 2>&1
cat $O/gs10.txt
timeout -k 10 300 python tools/pack_model.py 4 4096 > $O/pack_model_c4.txt 2>&1; tail -8 $O/pack_model_c4.txt
timeout -k 10 300 python tools/pack_model.py 8 4096 > $O/pack_model_c8.txt 2>&1; tail -8 $O/pack_model_c8.txt
