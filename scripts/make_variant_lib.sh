#!/bin/bash
# Experiment build of the library with extra compiler defines:
#   scripts/make_variant_lib.sh <name> "<-D...>"  ->  scripts/bin/libpyrad_hip_<name>.so
# Select it with PYRAD_HIP_LIB=<path>.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; DEFS=$2
mkdir -p $ROOT/scripts/bin
T=$(mktemp -d)
mkdir -p $T/pyrad_amd/csrc $T/include
cp $ROOT/pyrad_amd/csrc/* $T/pyrad_amd/csrc/
cp $ROOT/include/pyrad_hip.h $T/include/
make -C $T/pyrad_amd/csrc -j4 EXTRA="$DEFS" > $T/build.log 2>&1 || { grep -E "error" $T/build.log; exit 1; }
cp $T/pyrad_amd/lib/libpyrad_hip.so $ROOT/scripts/bin/libpyrad_hip_$NAME.so
rm -rf $T
echo built scripts/bin/libpyrad_hip_$NAME.so
