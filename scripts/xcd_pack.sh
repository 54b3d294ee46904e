R=$GRAFT_REPO_ROOT
cd $R && timeout -k 10 500 python -m pytest tests/test_gpu_schedule.py tests/test_gpu_merged.py -q -x -m gpu > gpurun_out/t1.log 2>&1 || { tail -30 gpurun_out/t1.log; exit 1; }
tail -2 gpurun_out/t1.log
for args in "--workload C2" "--workload C3 --shard-of 8,4 --step per-list" "--workload C3 --shard-of 8,2 --step per-list"; do
for c in 1 0 t-1 t6 t10; do tol=""; case $c in t*) tol="--set accum_xcd_tolerance=${c#t}"; esac; for ch in $(echo 0); do
  bash $R/scripts/quick_bench.sh "[$args] xcd_pack=$c chunks=$ch" $args $( case $c in t*) echo "--set accum_xcd_pack=1";; *) echo "--set accum_xcd_pack=$c";; esac ) $tol
  OUT=$R/gpurun_out/xp_$c; rm -rf $OUT
  (cd /tmp && TMPDIR=/tmp rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT -- python3 $R/bench.py --steps 6 --warmup 2 --blocks 1 --no-cpu-baseline --no-api-path --no-direct-pass $args $( case $c in t*) echo "--set accum_xcd_pack=1";; *) echo "--set accum_xcd_pack=$c";; esac ) $tol > $OUT.log 2>&1)
  python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    tot = collections.defaultdict(float); n = collections.defaultdict(set)
    for row in csv.DictReader(open(f)):
        if "accumulate" in row["Kernel_Name"]:
            tot[row["Counter_Name"]] += float(row["Counter_Value"]); n[row["Counter_Name"]].add(row["Dispatch_Id"])
    for c, v in tot.items(): print("   ", c, "x2 per launch: %.1f MB" % (2 * v * 1024 / len(n[c]) / 1e6))
PY
done; done; done
bash $R/scripts/pack_time.sh c3s8 --workload C3 --shard-of 8,2
