#!/bin/bash
# Experiment build from a PATCHED copy of the sources (the tree stays untouched):
#   scripts/make_patched_lib.sh <name> <python-file-that-edits-the-copy> ["<-D...>"]
# The python file is run with the copy's csrc directory as argv[1].  -> scripts/bin/libpyrad_hip_<name>.so
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; PATCH=$2; DEFS=$3
mkdir -p $ROOT/scripts/bin
T=$(mktemp -d)
mkdir -p $T/pyrad_amd/csrc $T/include
cp $ROOT/pyrad_amd/csrc/* $T/pyrad_amd/csrc/
cp $ROOT/include/pyrad_hip.h $T/include/
python3 $PATCH $T/pyrad_amd/csrc
make -C $T/pyrad_amd/csrc -j4 EXTRA="$DEFS" > $T/build.log 2>&1 || { grep -E "error" $T/build.log; exit 1; }
cp $T/pyrad_amd/lib/libpyrad_hip.so $ROOT/scripts/bin/libpyrad_hip_$NAME.so
rm -rf $T
echo built scripts/bin/libpyrad_hip_$NAME.so
