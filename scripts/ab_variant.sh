# same-box check of an experiment build against the production library: bit-identity on C1, C2, C3, the column, then alternating step times
# usage (GPU box): ab_variant.sh <lib name under scripts/bin without the libpyrad_hip_ prefix>
cd $GRAFT_REPO_ROOT
V=$1
bash scripts/bitcmp_libs.sh prod $V | tail -2
bash scripts/ab_libs.sh "prod $V" --workload C3
bash scripts/ab_libs.sh "prod $V" --workload C5 | cut -c1-200
bash scripts/ab_libs.sh "prod $V" --workload C3 --shard-of 8,4
bash scripts/ab_libs.sh "prod $V" --workload C2
