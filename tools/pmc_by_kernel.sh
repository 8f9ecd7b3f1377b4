#!/bin/bash
# Instruction counts per kernel: the split execution (step kernel without the sweeps + arb_gsw_kernel = the sweeps alone)
# and the contact-free workload, to apportion the fused kernel's VALU instructions between phases.
# usage (GPU box): tools/pmc_by_kernel.sh <tag>   -> gpurun_out/<tag>/{split,nocontact}
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-pmck}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--steps 40 --warmup 40 --min-seconds 0.2 --no-cpu-baseline --no-per-step-leg"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/split -- python3 $R/bench.py $B --split wave > $OUT/split.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/nocontact -- python3 $R/bench.py $B --contacts 0 > $OUT/nocontact.log 2>&1
python3 - <<PY
import csv, glob, collections
for leg in ("split", "nocontact"):
    f = glob.glob("$OUT/%s/*/*counter_collection.csv" % leg)
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        k = r["Kernel_Name"].split("<")[0].split("(")[0][-40:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_INSTS_VALU": cnt[k] += 1
    for k in acc:
        print(leg, k, "dispatches", cnt[k], {c: round(v / max(cnt[k], 1)) for c, v in acc[k].items()})
PY
