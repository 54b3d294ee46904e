# bit-for-bit comparison of two library builds on C1, C2, C3 (merged and per-list) and the column: usage bitcmp_libs.sh <lib a> <lib b>
# (names under scripts/bin without the libpyrad_hip_ prefix, or 'prod')
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for l in $1 $2; do
  if [ "$l" = prod ]; then unset PYRAD_HIP_LIB; else export PYRAD_HIP_LIB=$R/scripts/bin/libpyrad_hip_$l.so; fi
  python3 $R/scripts/dump_spectra.py /tmp/dump_$l > /dev/null || exit 1
done
python3 - /tmp/dump_$1 /tmp/dump_$2 <<'PY'
import sys, os, numpy as np
a, b = sys.argv[1:3]
bad = 0
for f in sorted(os.listdir(a)):
    x, y = np.load(os.path.join(a, f)), np.load(os.path.join(b, f))
    same = x.shape == y.shape and np.array_equal(x, y, equal_nan=True)
    bad += 0 if same else 1
    print(f, "identical" if same else "DIFFERENT: max rel %.3e at %d points" % (np.max(np.abs(x - y) / np.maximum(np.abs(y), 1e-300)), np.count_nonzero(x != y)))
print("bit-identical" if bad == 0 else "%d files differ" % bad)
PY
