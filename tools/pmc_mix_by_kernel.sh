#!/bin/bash
# Dynamic instruction mix per kernel (round 4): the split execution separates the Gauss-Seidel sweeps (arb_gsw_kernel) from
# everything else (arb_step_kernel), per world-step of the 4096-world x 40-step headline workload.
# usage (GPU box): tools/pmc_mix_by_kernel.sh <tag>   -> gpurun_out/<tag>/mix_split*.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-pmcmix}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="--steps 40 --warmup 40 --min-seconds 0.2 --no-cpu-baseline --no-per-step-leg --split wave"
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $OUT/m1 -- python3 $R/bench.py $B > $OUT/m1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d $OUT/m2 -- python3 $R/bench.py $B > $OUT/m2.log 2>&1
python3 - <<PY > $OUT/mix_split.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); eps = collections.Counter()
for leg in ("m1", "m2"):
    f = glob.glob("$OUT/%s/*/*counter_collection.csv" % leg)
    for r in csv.DictReader(open(f[0])):
        k = "gsw" if "arb_gsw" in r["Kernel_Name"] else "step" if "arb_step_kernel" in r["Kernel_Name"] else None
        if k is None: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_FMA_F32"): eps[(k, leg)] += 1
for k in acc:
    for c, v in sorted(acc[k].items()):
        leg = "m2" if c in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_ACTIVE_INST_VALU") else "m1"
        n = eps[(k, leg)]          # dispatches of this kernel in that pass; one dispatch = one step of 4096 worlds (gsw) / 4096 worlds (step: 41 per episode)
        print("%-5s %-28s %12.1f per dispatch per world" % (k, c, v / max(n, 1) / 4096.))
PY
cat $OUT/mix_split.txt
