#!/bin/bash
# how long the device-built schedule's kernels take (first use of a window): rocprofv3 kernel stats of one short bench run
# usage (GPU box): bash scripts/pack_time.sh <label> <bench args...>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
label=$1; shift
OUT=$R/gpurun_out/pack_$label
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 5 --warmup 2 --blocks 1 --no-cpu-baseline --no-api-path --no-direct-pass "$@" > $OUT/run.log 2>&1
f=$(find $OUT -name "*kernel_stats.csv" | head -1)
grep -E "sched_|merge_rank|centre_index|span_table" "$f" | cut -c1-160
