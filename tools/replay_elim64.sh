#!/bin/bash
# Run on the GPU box: the float64-elimination experiment (build/ab/r4elim64.so = tools/quick_build.sh r4elim64 -DARB_ELIM_F64=1
# with ARB_QUICK=2): phases C + D of float32 worlds in float64, float32 sweeps -- outlier rate on the seeds of the default
# sample.  -> gpurun_out/<tag>/replay_elim64.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-replay}; mkdir -p $OUT
: > $OUT/replay_elim64.txt
for spec in "1000 4 1 4" "7 4 1 4" "1000 8 1 8"; do
  echo "== ARB_ELIM_F64: tools/replay_stats.py $spec  (seed, world stride, step stride, contacts)" >> $OUT/replay_elim64.txt
  ARBSTEP_LIB=$R/build/ab/r4elim64.so python3 $R/tools/replay_stats.py $spec >> $OUT/replay_elim64.txt 2>&1 || exit 1
done
