#!/bin/bash
# Diagnostic build of the library (scripts/bin/libpyrad_hip_diag.so): -DLBL_DIAG compiles the timing-only
# ablations (lbl_set_option "debug_ablate": parts of kernels switched off, WRONG results) and the LBL_DIAG_*
# environment knobs into a COPY of the sources.  The production library (pyrad_amd/lib) carries none of that.
#   PYRAD_HIP_LIB=$PWD/scripts/bin/libpyrad_hip_diag.so python bench.py --set debug_ablate=1 ...
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
T=$(mktemp -d)
mkdir -p $T/pyrad_amd/csrc $T/include $ROOT/scripts/bin
cp $ROOT/pyrad_amd/csrc/* $T/pyrad_amd/csrc/
cp $ROOT/include/pyrad_hip.h $T/include/
make -C $T/pyrad_amd/csrc -j4 EXTRA=-DLBL_DIAG > $T/build.log 2>&1 || { grep -E "error" $T/build.log; exit 1; }
cp $T/pyrad_amd/lib/libpyrad_hip.so $ROOT/scripts/bin/libpyrad_hip_diag.so
rm -rf $T
echo built scripts/bin/libpyrad_hip_diag.so
