# kernel-trace of one bench invocation; prints per-kernel average durations
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/trace_tmp; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline $BENCH_ARGS > $OUT/log.txt 2>&1
cat $OUT/*/*kernel_stats.csv | cut -c1-160
