#!/bin/bash
# SQ counters of bench.py (per-step launches, 40 steps) for each development library given.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  export ARBSTEP_LIB=$R/$lib
  tag=$(basename $lib .so)
  OUT=$R/gpurun_out/abpmc_$tag
  mkdir -p $OUT
  B="python3 $R/bench.py --steps 40 --warmup 40 --no-cpu-baseline --steps-per-launch 40"
  timeout 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/a -- $B > $OUT/a.log 2>&1
  timeout 150 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC --output-format csv -d $OUT/b -- $B > $OUT/b.log 2>&1
  python3 - $OUT $tag <<'P'
import csv, glob, sys
out, tag = sys.argv[1], sys.argv[2]
acc = {}
for f in glob.glob(out + "/*/*/*_counter_collection.csv"):
    rows = [r for r in csv.DictReader(open(f)) if "arb_step_kernel" in r["Kernel_Name"]]
    ids = sorted(set(int(r["Dispatch_Id"]) for r in rows))[:2]     # the two 40-step episode launches
    for r in rows:
        if int(r["Dispatch_Id"]) in ids:
            acc[r["Counter_Name"]] = acc.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"]) / len(ids)
ws = 4096 * 40.0
print(tag, " ".join("%s=%.1f" % (k.replace("SQ_", ""), v / ws) for k, v in sorted(acc.items())), "(per world-step)")
P
done
