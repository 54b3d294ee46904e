# kernel trace + SQ passes of the bench step (run on the GPU box): BENCH_ARGS picks workload / step / options
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-m}
OUT=$R/gpurun_out/pmc_$TAG
mkdir -p $OUT
ARGS="$R/bench.py --steps 10 --warmup 2 --blocks 1 --no-cpu-baseline --no-api-path --no-direct-pass ${BENCH_ARGS:---workload C3}"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ARGS > $OUT/trace.log 2>&1 || tail -5 $OUT/trace.log
pass() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $ARGS > $OUT/$name.log 2>&1 || tail -5 $OUT/$name.log; }
pass p1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD
pass p2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        print("stats", row["Name"][:70], row["Calls"], row["AverageNs"], row["Percentage"])
for d in sorted(glob.glob("$OUT/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        disp = collections.defaultdict(set)
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:60]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); disp[k].add(row["Dispatch_Id"])
        for k, v in agg.items():
            if "accumulate" in k or "sweep" in k or "prep" in k or "column" in k:
                print(k, len(disp[k]), {c: "%.5g" % (x / len(disp[k])) for c, x in v.items()})
PY
