#!/bin/bash
# Run on the GPU box: instruction counters of the headline launch for several development libraries (build/ab/<name>.so).
# usage: tools/pmc_ab.sh <out-tag> <lib> [<lib> ...]   -> gpurun_out/<out-tag>/pmc_ab.txt (per world-step)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift; mkdir -p $OUT; : > $OUT/pmc_ab.txt
cd /tmp && export TMPDIR=/tmp
B="--steps 40 --warmup 40 --min-seconds 0.2 --no-cpu-baseline --no-per-step-leg"
for lib in "$@"; do
  export ARBSTEP_LIB=$R/build/ab/$lib.so
  rm -rf $OUT/p1 $OUT/p2
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -- python3 $R/bench.py $B > $OUT/p1.log 2>&1
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH SQ_WAIT_INST_LDS --output-format csv -d $OUT/p2 -- python3 $R/bench.py $B > $OUT/p2.log 2>&1
  python3 - <<PY >> $OUT/pmc_ab.txt
import csv, glob, collections
acc = collections.defaultdict(float); n = collections.Counter()
for leg in ("p1", "p2"):
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % leg):
        for r in csv.DictReader(open(f)):
            if "arb_step_kernel" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print("$lib: " + "  ".join("%s %.0f" % (k.replace("SQ_", ""), acc[k] / n[k] / (4096 * 40.)) for k in sorted(acc)))
PY
done
cat $OUT/pmc_ab.txt
