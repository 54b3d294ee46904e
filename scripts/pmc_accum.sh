# PMC passes on the accumulate kernel (run on the GPU box).  Output: gpurun_out/pmc/*.csv
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
ARGS="$R/bench.py --steps 3 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:---workload C3 --points-per-lane 4 --line-split 1}"
pass() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $OUT/$name -- python3 $ARGS > $OUT/$name.log 2>&1 || tail -5 $OUT/$name.log; }
pass p1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD
pass p2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY
pass p3 GRBM_GUI_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/p*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"][:60]
            agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); 
        for k, v in agg.items():
            if "accumulate" in k or "sweep" in k or "prep" in k:
                print(k, {c: "%.4g" % x for c, x in v.items()})
PY
