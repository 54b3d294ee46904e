R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2 3; do
for c in 1 0; do
  OUT=$R/gpurun_out/f8_$c; rm -rf $OUT
  rocprofv3 --pmc ${CTR:-FETCH_SIZE} --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 20 --warmup 3 --blocks 1 --no-cpu-baseline --no-api-path --no-direct-pass --workload C3 --shard-of 8,4 --set accum_xcd_pack=$c > $OUT.log 2>&1
  python3 - "$OUT" "$c" <<'PY'
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    tot = collections.defaultdict(float); n = collections.defaultdict(set)
    for row in csv.DictReader(open(f)):
        if "accumulate" in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]].add(row["Dispatch_Id"])
    for c, v in tot.items(): print("xcd_pack", sys.argv[2], c, "raw KB-units per launch: %.2f MB-equivalent (x1) over %d launches" % (v * 1024 / len(n[c]) / 1e6, len(n[c])))
PY
done; done
