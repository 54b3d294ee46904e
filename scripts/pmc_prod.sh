# instruction counts and wave-cycle buckets of the production library's accumulate kernel (GPU box): pmc_prod.sh [bench args]
R=$GRAFT_REPO_ROOT
BENCH_ARGS="${*:---workload C3}" bash $R/scripts/pmc_merged.sh prod 2>&1 | grep -E "accumulate" | cut -c1-420
