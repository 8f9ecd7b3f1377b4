#!/bin/bash
# Run on the GPU box: the 160k-world-step adjudication sample (seeds 1000 / 7 / 3 with 4 contacts, every 4th world, every
# step; seeds 1000 / 7 with 8 contacts, every 8th world) -> gpurun_out/<tag>/replay_stats.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/${1:-replay}; mkdir -p $OUT
: > $OUT/replay_stats.txt
for spec in "1000 4 1 4" "7 4 1 4" "3 4 1 4" "1000 8 1 8" "7 8 1 8"; do
  echo "== tools/replay_stats.py $spec  (seed, world stride, step stride, contacts)" >> $OUT/replay_stats.txt
  python3 $R/tools/replay_stats.py $spec >> $OUT/replay_stats.txt 2>&1 || exit 1
done
