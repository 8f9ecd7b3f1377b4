#!/bin/bash
# GPU box: bit-identity (tools/xlib_dump.py) and A/B of two development builds.  usage: fast_ab.sh <base> <new>
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/fast; mkdir -p $O; L=$O/ab_$2.txt; : > $L
cd $R
ARBSTEP_LIB=$R/build/ab/$1.so timeout -k 10 200 python3 tools/xlib_dump.py $O/a.npz quick >> $L 2>&1 || exit 1
ARBSTEP_LIB=$R/build/ab/$2.so timeout -k 10 200 python3 tools/xlib_dump.py $O/b.npz quick >> $L 2>&1 || exit 1
python3 tools/xlib_cmp.py $O/a.npz $O/b.npz >> $L 2>&1; rm -f $O/*.npz
B="python3 $R/bench.py --steps 40 --warmup 40 --min-seconds 2 --no-cpu-baseline"
v() { python3 -c "
import sys,json
j=json.loads(sys.stdin.readline())
print('%.3f M  per-step %.3f M' % (j['value']/1e6, (j.get('per_step_launch',{}).get('value') or 0)/1e6))"; }
w() { python3 -c "import sys,json; print('%.3f M' % (json.loads(sys.stdin.readline())['value']/1e6))"; }
for round in 1 2; do
  for lib in $1 $2; do
    echo "round $round $lib: 4096: $(ARBSTEP_LIB=$R/build/ab/$lib.so ARB_BENCH_LEGS=perstep $B 2>/dev/null | v)   65536: $(ARBSTEP_LIB=$R/build/ab/$lib.so $B --no-per-step-leg --batch 65536 2>/dev/null | w)  mpc 2048x32: $(ARBSTEP_LIB=$R/build/ab/$lib.so $B --no-per-step-leg --config 5 --batch 2048 --steps 32 --warmup 32 2>/dev/null | w)" >> $L || exit 1
  done
done
cat $L
