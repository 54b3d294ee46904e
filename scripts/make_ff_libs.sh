#!/bin/bash
# Experiment builds of the library with other far-field parameters (threshold in half-spans, series
# terms): scripts/bin/libpyrad_hip_ff<FAR>_<NT>.so.  Select one with PYRAD_HIP_LIB=<path>.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/scripts/bin
for cfg in "$@"; do
  FAR=${cfg%_*}; NT=${cfg#*_}
  T=$(mktemp -d)
  mkdir -p $T/pyrad_amd/csrc $T/include
  cp $ROOT/pyrad_amd/csrc/* $T/pyrad_amd/csrc/
  cp $ROOT/include/pyrad_hip.h $T/include/
  make -C $T/pyrad_amd/csrc -j4 EXTRA="-DLBL_FF_FAR=$FAR -DLBL_FF_NT=$NT" > $T/build.log 2>&1 || { grep -E "error" $T/build.log; exit 1; }
  cp $T/pyrad_amd/lib/libpyrad_hip.so $ROOT/scripts/bin/libpyrad_hip_ff${FAR}_${NT}.so
  rm -rf $T
  echo built scripts/bin/libpyrad_hip_ff${FAR}_${NT}.so
done
