#!/bin/bash
# Boxes of the pool differ by +-4 %: collect the profile set only on a box that is not one of the slow ones.
R=$GRAFT_REPO_ROOT
MS=$(python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-api-path --no-direct-pass | python3 -c "import sys,json; print(json.load(sys.stdin)['ms_per_step'])")
echo "C3 ms_per_step on this box: $MS"
python3 -c "import sys; sys.exit(0 if float('$MS') <= float('${LIMIT:-0.3145}') else 3)" || { echo "slow box: not collecting"; exit 0; }
for a in "C3 8,4" "C3 4,2" "C3 2,1" "C3" "C5" "C2" "C1"; do bash $R/profiles/collect.sh r03 $a > $R/gpurun_out/collect_$(echo $a | tr ' ,' '__').log 2>&1; echo "$a rc $?"; done
echo collected
