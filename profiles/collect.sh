#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box:
#   gpurun -- 'bash profiles/collect.sh r05 C3'     (workloads: C1 C2 C3 C5; a third argument G,r profiles one shard)
# writes gpurun_out/<tag>/..., which profiles/summarize.py turns into the committed summaries
# (gpurun_out/<tag>/final/: copy its files into profiles/).
# Counters are collected in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one TCC pass,
# MI355X_MICROARCH.md "rocprofv3 PMC slots") and never together with the trace domains.
set -e
TAG=${1:-r06}
WORKLOAD=${2:-C3}
SHARD=$3          # optional "G,r": shard r of a G-way sharding (what a rank of a G-GPU run computes); label <workload>s<G>
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
WL=$WORKLOAD
EXTRA=""
if [ -n "$SHARD" ]; then WL=${WORKLOAD}s${SHARD%%,*}; EXTRA="--shard-of $SHARD"; fi
# (--step is pinned to what --step auto picks for the shape, so that the committed profiles do not depend on a measured threshold:
#  advisor, round 5; --legs none: the other workloads have collections of their own)
STEP=merged
case "$WL" in C3s8|C3s4) STEP=per-list;; esac
CMD="$R/bench.py --steps 20 --warmup 3 --blocks 1 --no-cpu-baseline --no-api-path --no-direct-pass --legs none --step $STEP --workload $WORKLOAD $EXTRA"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$WL -- python3 $CMD > $OUT/trace_$WL.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch_$WL -- python3 $CMD > $OUT/pmc_fetch_$WL.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/pmc_write_$WL -- python3 $CMD > $OUT/pmc_write_$WL.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_valu_$WL -- python3 $CMD > $OUT/pmc_valu_$WL.log 2>&1
# round 5: where the accumulate kernel's wave-cycles go (MI355X_MICROARCH.md "rocprofv3 PMC slots": WAIT_ANY + WAIT_INST_ANY +
# ACTIVE_INST_ANY = WAVE_CYCLES, disjoint, in quad-cycles) and its instruction mix, in two more SQ passes of 8 counters
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $OUT/pmc_sq2_$WL -- python3 $CMD > $OUT/pmc_sq2_$WL.log 2>&1
rocprofv3 --pmc SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/pmc_sq3_$WL -- python3 $CMD > $OUT/pmc_sq3_$WL.log 2>&1
# summaries first (pmc_traffic.json then carries this build's source hash), then the bench line that quotes them
python3 $R/profiles/summarize.py $OUT $TAG $WL > $OUT/summary_$WL.txt
python3 $R/bench.py --steps 20 --warmup 3 --legs none --workload $WORKLOAD $EXTRA > $R/profiles/${TAG}_${WL}_bench.json 2> $OUT/bench_$WL.err
# which box was this?  every collection appends its line to the round's list (profiles/<tag>_boxes.jsonl): the set is collected
# ONCE per round, whatever box comes up; the steady-state blocks inside the bench line say how much this box scatters
python3 - "$R/profiles/${TAG}_${WL}_bench.json" "$R/profiles/${TAG}_boxes.jsonl" "$WL" <<'PY'
import json, sys, time, socket
d = json.load(open(sys.argv[1]))
rec = {"workload": sys.argv[3], "ms_per_step": d["ms_per_step"], "blocks": d.get("ms_per_step_blocks", {}).get("blocks"),
       "budget_ms_per_step": (d.get("budget_leg") or {}).get("ms_per_step"), "utc": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
       "host": socket.gethostname(), "source": "profiles/collect.sh"}
open(sys.argv[2], "a").write(json.dumps(rec) + "\n")
PY
# everything to commit, where gpurun brings it back from
mkdir -p $OUT/final
cp $R/profiles/${TAG}_${WL}_* $R/profiles/pmc_traffic.json $R/profiles/${TAG}_boxes.jsonl $OUT/final/
cat $R/profiles/${TAG}_${WL}_kernel_stats.csv
