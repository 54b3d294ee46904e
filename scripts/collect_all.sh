#!/bin/bash
# The round's profile set on whatever box comes up (no box is skipped or re-rolled), in two gpurun calls (seven rocprofv3 /
# bench passes per workload since round 5):
#   gpurun --timeout 1200 -- 'bash scripts/collect_all.sh r05 a'      (C3 and its shards of 8, 4, 2)
#   gpurun --timeout 1200 -- 'bash scripts/collect_all.sh r05 b'      (C5, C2, C1)
# Copy gpurun_out/<tag>/final/* into profiles/ BETWEEN the two calls: a box only receives the repo (gpurun_out/ does not travel),
# so the second call extends the profiles/pmc_traffic.json and profiles/<tag>_boxes.jsonl it finds there.
R=$GRAFT_REPO_ROOT
TAG=${1:-r06}
PART=${2:-a}
mkdir -p $R/gpurun_out
if [ "$PART" = a ]; then set -- "C3" "C3 8,4" "C3 4,2" "C3 2,1"; else set -- "C5" "C2" "C1"; fi
for a in "$@"; do
  bash $R/profiles/collect.sh $TAG $a > $R/gpurun_out/collect_$(echo $a | tr ' ,' '__').log 2>&1
  echo "$a rc $?"
done
echo collected
