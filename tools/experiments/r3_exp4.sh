#!/bin/bash
O=gpurun_out/r3h; mkdir -p $O
timeout -k 10 120 ./build/valu_issue_probe > $O/valu_probe.txt 2>&1; grep -E "independent|dependent" $O/valu_probe.txt | cut -c1-200
B="python bench.py --no-cpu-baseline --no-per-step-leg --min-seconds 1.5"
val() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"; }
for c in 2 3 4 5 8; do for t in 2 4 8; do echo -n "chunk $c tail $t: "; ARB_QUEUE_CHUNK=$c ARB_QUEUE_TAIL=$t timeout -k 10 100 $B 2>/dev/null | val; done; done | tee $O/queue_sweep.txt
echo -n "config5 chunk 4 tail 4: "; timeout -k 10 100 $B --config 5 2>/dev/null | val
echo -n "config5 chunk 8 tail 4: "; ARB_QUEUE_CHUNK=8 timeout -k 10 100 $B --config 5 2>/dev/null | val
