#!/bin/bash
# Instruction-cache counters of the bench workload (one pass).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/icache
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $R/bench.py --steps 40 --warmup 40 --no-cpu-baseline"
timeout 150 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES --output-format csv -d $OUT/a -- $B > $OUT/a.log 2>&1
python3 - $OUT <<'P'
import csv, glob, sys
acc = {}
for f in glob.glob(sys.argv[1] + "/a/*/*_counter_collection.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "arb_step_kernel" in r["Kernel_Name"]]
    ids = sorted(set(int(r["Dispatch_Id"]) for r in rows))[:2]
    for r in rows:
        if int(r["Dispatch_Id"]) in ids:
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"]) / len(ids)
ws = 4096 * 40.0
print(" ".join("%s=%.1f" % (k, v / ws) for k, v in sorted(acc.items())), "(per world-step)")
P
tail -2 $OUT/a.log | cut -c1-200
