# same-box A/B of library builds: usage ab_libs.sh "<lib names under scripts/bin without prefix, or 'prod'>" <bench args...>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
LIBS=$1; shift
for rep in 1 2; do
for l in $LIBS; do
  if [ "$l" = prod ]; then unset PYRAD_HIP_LIB; else export PYRAD_HIP_LIB=$R/scripts/bin/libpyrad_hip_$l.so; fi
  bash $R/scripts/quick_bench.sh "$l" "$@"
done
done
