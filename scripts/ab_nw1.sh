cd $GRAFT_REPO_ROOT
bash scripts/bitcmp_libs.sh prod nw1 | tail -3
bash scripts/ab_libs.sh "prod nw1" --workload C3
bash scripts/ab_libs.sh "prod nw1" --workload C3 --step per-list
bash scripts/ab_libs.sh "prod nw1" --workload C5 | cut -c1-200
bash scripts/ab_libs.sh "prod nw1" --workload C3 --shard-of 8,4
bash scripts/ab_libs.sh "prod nw1" --workload C3 --shard-of 2,1
