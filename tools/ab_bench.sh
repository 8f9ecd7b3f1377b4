#!/bin/bash
# Same-box A/B of development builds (tools/quick_build.sh): AB_VARIANTS="a b" [AB_ARGS="--batch 65536"] tools/ab_bench.sh
# runs bench.py on build/ab/<variant>.so twice, interleaved, and prints world-steps/s.
set -e
for r in 1 2; do
 for v in ${AB_VARIANTS:-old new}; do
  echo -n "$v: "; ARBSTEP_LIB=build/ab/$v.so timeout -k 10 120 python bench.py --no-cpu-baseline --no-per-step-leg --min-seconds 2 ${AB_ARGS} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'])"
 done
done
