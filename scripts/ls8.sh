R=$GRAFT_REPO_ROOT
for w in "8,4" "4,2" "2,1"; do
for ls in 0 4 8; do
  bash $R/scripts/quick_bench.sh "s$w merged LS=$ls" --workload C3 --shard-of $w --step merged --set accum_line_split=$ls
done; done
bash $R/scripts/quick_bench.sh "C2 LS=0" --workload C2
bash $R/scripts/quick_bench.sh "C2 LS=8" --workload C2 --set accum_line_split=8
bash $R/scripts/quick_bench.sh "C2 LS=4" --workload C2 --set accum_line_split=4
bash $R/scripts/quick_bench.sh "s8 per-list LS=8" --workload C3 --shard-of 8,4 --step per-list --set accum_line_split=8
bash $R/scripts/quick_bench.sh "s8 per-list LS=0" --workload C3 --shard-of 8,4 --step per-list
