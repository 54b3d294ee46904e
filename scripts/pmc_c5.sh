#!/bin/bash
# SQ counters of the column step's kernels:  gpurun -- 'bash scripts/pmc_c5.sh "accum_skew=1" "accum_skew=0"'
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_c5
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for opts in "$@"; do
  i=$((i+1))
  SET=""
  for o in $opts; do SET="$SET --set $o"; done
  rm -rf $OUT/p$i; rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/bench.py --workload ${WORKLOAD:-C5} --steps 3 --warmup 1 --precondition-seconds 0 --no-cpu-baseline --no-api-path --no-direct-pass $SET > $OUT/p$i.json 2> $OUT/p$i.err || exit 1
  echo "== $opts"
  python3 - $OUT/p$i <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:50]
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        key = (r['Dispatch_Id'])
        if key not in seen: seen.add(key); n[k] += 1
for k in acc:
    print(k, 'launches', n[k], {c: '%.4g' % (v / n[k]) for c, v in acc[k].items()})
PY
done
