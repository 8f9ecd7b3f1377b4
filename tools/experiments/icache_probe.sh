#!/bin/bash
# Run on the GPU box: does the step kernel (~190 KB of straight-line code, every wave in its own phase) miss the
# instruction cache?  Lists the instruction-cache / fetch counters this rocprofv3 knows and collects them on the bench.
# usage: tools/icache_probe.sh <tag>  -> gpurun_out/<tag>/
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-icache}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -i -E "icache|ifetch|inst_cache|SQC_|SQ_INST_LEVEL|SQ_WAIT_INST|SQ_INSTS_BRANCH|SQ_IFETCH" $OUT/counters.txt | cut -c1-200 | sort -u | head -80 > $OUT/counters_icache.txt
BENCH_S="python3 $R/bench.py --steps 40 --warmup 40 --min-seconds 0.2 --no-cpu-baseline --no-per-step-leg"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE --output-format csv -d $OUT/pmc_ic1 -- $BENCH_S > $OUT/pmc_ic1.log 2>&1
rocprofv3 --pmc SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/pmc_ic2 -- $BENCH_S > $OUT/pmc_ic2.log 2>&1
rocprofv3 --pmc SQC_ICACHE_INPUT_VALID_READYB SQC_TC_INST_REQ SQC_TC_REQ SQC_ICACHE_BUSY_CYCLES --output-format csv -d $OUT/pmc_ic3 -- $BENCH_S > $OUT/pmc_ic3.log 2>&1
for f in $OUT/*.log; do echo "== $f"; tail -3 $f | cut -c1-300; done
find $OUT -name "*counter_collection.csv" | head
