# VALU / SALU / LDS instruction counts of the accumulate kernel per launch with parts switched off (diagnostic library:
# scripts/make_diag_lib.sh first).  usage: pmc_ablate.sh <tag> "<bits ...>" <bench args...>      (run on the GPU box)
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=$1; BITS=$2; shift; shift
export PYRAD_HIP_LIB=$R/scripts/bin/libpyrad_hip_diag.so
for b in $BITS; do
  OUT=$R/gpurun_out/pmcab_${TAG}_$b
  rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $OUT -- \
    python3 $R/bench.py --steps 6 --warmup 2 --blocks 1 --no-cpu-baseline --no-api-path --no-direct-pass --legs none "$@" --set debug_ablate=$b > $OUT/log.txt 2>&1 || tail -3 $OUT/log.txt
  python3 - "$OUT" "$TAG" "$b" <<'PY'
import csv, glob, collections, sys
out, tag, bits = sys.argv[1:4]
for f in glob.glob(out + "/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
    dur = collections.defaultdict(list)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"][:48]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"]); disp[k].add(row["Dispatch_Id"])
    for k, v in agg.items():
        if "accumulate" in k:
            n = len(disp[k])
            print(tag, "ablate", bits, k, n, {c: "%.4g" % (x / n) for c, x in sorted(v.items())})
PY
done
