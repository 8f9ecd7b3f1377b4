#!/bin/bash
# LDS bank conflicts of the headline launch per build (GPU box): SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per arb_step_kernel dispatch.
# usage: tools/pmc_lds.sh <out-tag> [--contacts N] <build> ...    (builds under build/ab, or "shipped")
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
EXTRA=""
if [ "$1" = "--contacts" ]; then EXTRA="--contacts $2"; shift 2; fi
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_$lib -- python3 $R/tools/ab_r5.py $EXTRA --rounds 1 --episodes 6 $lib > $OUT/pmc_$lib.log 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/pmc_$lib/*/*counter_collection.csv")
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_INSTS_LDS": cnt[k] += 1
for k in acc:
    a = acc[k]
    print("$lib", k, "dispatches", cnt[k], {c: "%.4g" % (v / max(cnt[k], 1)) for c, v in a.items()},
          "conflict/active %.3f" % (a["SQ_LDS_BANK_CONFLICT"] / max(a["SQ_LDS_IDX_ACTIVE"], 1)))
PY
done 2>&1 | tee $OUT/summary.txt
