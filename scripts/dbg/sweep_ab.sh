#!/bin/bash
# A/B of the sweep kernels' grid size on one box: gpurun -- 'bash scripts/dbg/sweep_ab.sh'
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cap in 4096 9375 4096 2048 3125 4688 9375; do
  export LBL_DIAG_SWEEP_BLOCKS=$cap
  rm -rf /tmp/sw_$cap
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sw_$cap -- python3 $R/bench.py --workload ${WORKLOAD:-C3} --steps 30 --warmup 3 --no-cpu-baseline --no-api-path --no-direct-pass > /dev/null 2>&1
  echo "cap $cap: $(python3 - /tmp/sw_$cap <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'sweep' in r['Name'] or 'column_step' in r['Name'] or 'lds_kernel' in r['Name']:
            print(r['Name'][:34], round(float(r['AverageNs'])/1e3, 2), end='; ')
PY
)"
done
