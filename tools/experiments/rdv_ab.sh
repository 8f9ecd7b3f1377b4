#!/bin/bash
# GPU box: the rendezvous build (ARB_FORCE_RDV=1, build/ab/r4rdv.so) against the default selection, and the cost of
# single-step queue items alone (ARB_QUEUE_CHUNK=1 ARB_QUEUE_TAIL=0) -- round 4.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/rdv; mkdir -p $O; L=$O/ab.txt; : > $L
export ARBSTEP_LIB=$R/build/ab/r4rdv.so
B="python3 $R/bench.py --steps 40 --warmup 40 --min-seconds 2 --no-cpu-baseline --no-per-step-leg"
v() { python3 -c "import sys,json; print('%.3f' % (json.loads(sys.stdin.readline())['value']/1e6))"; }
for round in 1 2; do
  echo "round $round: chunk 1 / tail 0, 4096: $(ARB_FORCE_RDV=0 ARB_QUEUE_CHUNK=1 ARB_QUEUE_TAIL=0 $B 2>/dev/null | v)  65536: $(ARB_FORCE_RDV=0 ARB_QUEUE_CHUNK=1 ARB_QUEUE_TAIL=0 $B --batch 65536 2>/dev/null | v)" >> $L || exit 1
  echo "round $round: default 16384: $(ARB_FORCE_RDV=0 $B --batch 16384 2>/dev/null | v)  rdv 16384: $(ARB_FORCE_RDV=1 $B --batch 16384 2>/dev/null | v)" >> $L || exit 1
  echo "round $round: config 5 default: $(ARB_FORCE_RDV=0 $B --config 5 --steps 32 --warmup 32 2>/dev/null | v)  rdv: $(ARB_FORCE_RDV=1 $B --config 5 --steps 32 --warmup 32 2>/dev/null | v)" >> $L || exit 1
done
cat $L
