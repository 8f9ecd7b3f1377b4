#!/bin/bash
# Run on the GPU box (round 6): the 160k-world-step adjudication sample of tools/replay_all.sh with the library's defaults, and
# the 4-contact seeds once more with ARB_STEP_BODY_COLUMNS (decide the default on data).
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-r6_replay}; mkdir -p $OUT
: > $OUT/replay_stats.txt
for spec in "1000 4 1 4" "7 4 1 4" "3 4 1 4" "1000 8 1 8" "7 8 1 8" "1000 4 1 4 body" "7 4 1 4 body" "3 4 1 4 body"; do
  echo "== tools/replay_stats.py $spec  (seed, world stride, step stride, contacts[, kernels])" >> $OUT/replay_stats.txt
  timeout -k 10 400 python3 $R/tools/replay_stats.py $spec >> $OUT/replay_stats.txt 2>&1 || exit 1
  tail -4 $OUT/replay_stats.txt
done
