#!/bin/bash
# rocprofv3 kernel stats of the column step for a few option sets:  gpurun -- 'bash scripts/prof_c5.sh "accum_skew=0" "accum_skew_points_per_lane=8"'
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_c5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for opts in "$@"; do
  i=$((i+1))
  SET=""
  for o in $opts; do SET="$SET --set $o"; done
  rm -rf $OUT/t$i; rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t$i -- python3 $R/bench.py --workload ${WORKLOAD:-C5} --steps 10 --warmup 2 --no-cpu-baseline --no-api-path --no-direct-pass $SET > $OUT/t$i.json 2> $OUT/t$i.err || exit 1
  echo "== $opts"
  python3 - $OUT/t$i <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        print("%-60s calls %5s avg_us %10.1f" % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
