R=$GRAFT_REPO_ROOT
for args in "--workload C2" "--workload C3 --shard-of 16,7 --step per-list" "--workload C3 --shard-of 16,7 --step merged" "--workload C3 --shard-of 8,4 --step per-list" "--workload C3 --shard-of 12,4 --step per-list" "--workload C2 --shard-of 2,0" "--workload C2 --set accum_line_split=1"; do
for c in 1 2 3 0; do
  bash $R/scripts/quick_bench.sh "[$args] xcd_pack=$c" $args --set accum_xcd_pack=$c --set accum_xcd_tolerance=-1
done; done
