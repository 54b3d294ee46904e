R=$GRAFT_REPO_ROOT
for l in prod reach; do
  if [ "$l" = prod ]; then unset PYRAD_HIP_LIB; else export PYRAD_HIP_LIB=$R/scripts/bin/libpyrad_hip_$l.so; fi
  echo "== $l"
  BENCH_ARGS="--workload C3" bash $R/scripts/pmc_merged.sh v_$l 2>&1 | grep -E "accumulate" | cut -c1-400
done
