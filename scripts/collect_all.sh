#!/bin/bash
# The round's whole profile set in one gpurun call, on whatever box comes up (no box is skipped or re-rolled):
#   gpurun --timeout 1200 -- 'bash scripts/collect_all.sh r04'
R=$GRAFT_REPO_ROOT
TAG=${1:-r04}
mkdir -p $R/gpurun_out
for a in "C3" "C3 8,4" "C3 4,2" "C3 2,1" "C5" "C2" "C1"; do
  bash $R/profiles/collect.sh $TAG $a > $R/gpurun_out/collect_$(echo $a | tr ' ,' '__').log 2>&1
  echo "$a rc $?"
done
echo collected
