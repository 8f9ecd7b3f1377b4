#!/bin/bash
# GPU box: all-roots route of the sliding solve (build/ab/r4robc.so) against the eigenvalue routine alone (r4rob0.so, -DARB_ROOT_ROBUST=0) -- round 4.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/rob; mkdir -p $O; L=$O/ab.txt; : > $L
cd $R
ARBSTEP_LIB=$R/build/ab/r4robc.so timeout -k 10 300 python3 -m pytest tests/test_gpu_device_solve.py -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $L; tail -15 $O/pytest.txt >> $L
ARBSTEP_LIB=$R/build/ab/r4rob0.so timeout -k 10 200 python3 tools/xlib_dump.py $O/a.npz quick >> $L 2>&1 || exit 1
ARBSTEP_LIB=$R/build/ab/r4robc.so timeout -k 10 200 python3 tools/xlib_dump.py $O/b.npz quick >> $L 2>&1 || exit 1
python3 tools/xlib_cmp.py $O/a.npz $O/b.npz >> $L 2>&1; rm -f $O/*.npz
B="python3 $R/bench.py --steps 40 --warmup 40 --min-seconds 2 --no-cpu-baseline"
v() { python3 -c "
import sys,json
j=json.loads(sys.stdin.readline())
print('%.3f M  per-step %s' % (j['value']/1e6, j.get('per_step_launch',{}).get('value')))"; }
for round in 1 2; do
  for lib in r4rob0 r4robc; do
    echo "round $round $lib: 4096: $(ARBSTEP_LIB=$R/build/ab/$lib.so ARB_BENCH_LEGS=perstep $B 2>/dev/null | v)   65536: $(ARBSTEP_LIB=$R/build/ab/$lib.so $B --no-per-step-leg --batch 65536 2>/dev/null | v)  mpc 2048x32: $(ARBSTEP_LIB=$R/build/ab/$lib.so $B --no-per-step-leg --config 5 --batch 2048 --steps 32 --warmup 32 2>/dev/null | v)" >> $L || exit 1
  done
done
ARBSTEP_LIB=$R/build/ab/r4robc.so timeout -k 10 200 python3 tools/fallback_probe.py 2>&1 | tail -12 >> $L
cat $L
