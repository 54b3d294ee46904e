# XCD chunk count sweep: step time and FETCH_SIZE of the accumulate kernel (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for args in "--workload C3" "--workload C3 --shard-of 8,4"; do
for c in 32 16 10 6 3 1; do
  bash $R/scripts/quick_bench.sh "[$args] xcd_chunks=$c" $args --set accum_xcd_chunks=$c
  OUT=$R/gpurun_out/xcd_$c; rm -rf $OUT
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 6 --warmup 2 --blocks 1 --no-cpu-baseline --no-api-path --no-direct-pass $args --set accum_xcd_chunks=$c > $OUT.log 2>&1
  python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    tot = collections.defaultdict(float); n = collections.defaultdict(set)
    for row in csv.DictReader(open(f)):
        if "accumulate" in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]].add(row["Dispatch_Id"])
    for c, v in tot.items(): print("   ", c, "x2 per launch: %.1f MB" % (2 * v * 1024 / len(n[c]) / 1e6))
PY
done; done
