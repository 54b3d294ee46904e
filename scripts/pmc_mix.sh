#!/bin/bash
# Instruction mix and stall counters of the kernels of one bench workload (separate rocprofv3 --pmc passes):
#   gpurun -- 'bash scripts/pmc_mix.sh C3'   ->  gpurun_out/pmc_mix_<workload>.txt
set -e
W=${1:-C3}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_mix_$W
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="$R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-api-path --no-direct-pass --workload $W"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" \
           "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -- python3 $CMD > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - $OUT <<'PY' > $R/gpurun_out/pmc_mix_$W.txt
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("lbl::", "")
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in agg:
    print(k)
    for c in sorted(agg[k]):
        print("   %-28s %16.0f per launch" % (c, agg[k][c] / max(len(n[k][c]), 1)))
PY
cat $R/gpurun_out/pmc_mix_$W.txt
