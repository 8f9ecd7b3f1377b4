#!/bin/bash
# round 3: float64 arithmetic in the Gauss-Seidel sweeps of float32 worlds (-DARB_GS_F64=1): outlier rate and throughput
O=gpurun_out/r3f; mkdir -p $O
for s in 1000 7; do
  timeout -k 10 500 python tools/replay_stats.py $s 4 1 > $O/replay_default_$s.txt 2>&1; tail -3 $O/replay_default_$s.txt
  ARBSTEP_LIB=build/ab/r3gsf64.so timeout -k 10 500 python tools/replay_stats.py $s 4 1 > $O/replay_gsf64_$s.txt 2>&1; tail -3 $O/replay_gsf64_$s.txt
done
B="python bench.py --no-cpu-baseline --no-per-step-leg --min-seconds 2"
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['roofline']['kernel_ms'])"; }
for r in 1 2; do
  echo -n "default: "; timeout -k 10 120 $B 2>/dev/null | val
  echo -n "gs f64: "; ARBSTEP_LIB=build/ab/r3gsf64.so timeout -k 10 120 $B 2>/dev/null | val
done | tee $O/bench.txt
ARBSTEP_LIB=build/ab/r3base.so python tools/xlib_dump.py $O/base.npz quick > /dev/null 2>&1; python tools/xlib_dump.py $O/cur.npz quick > /dev/null 2>&1; python tools/xlib_cmp.py $O/base.npz $O/cur.npz | tail -3
