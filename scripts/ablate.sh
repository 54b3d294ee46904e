#!/bin/bash
# Timing ablations of the far-field kernel (results are WRONG by construction; timing only):
# builds scripts/bin/libpyrad_hip_abl_<what>.so with one component of K2 removed.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $ROOT/scripts/bin
for what in nogauss nofar noedge nonear; do
  T=$(mktemp -d)
  mkdir -p $T/pyrad_amd/csrc $T/include
  cp $ROOT/pyrad_amd/csrc/* $T/pyrad_amd/csrc/
  cp $ROOT/include/pyrad_hip.h $T/include/
  python3 - "$T/pyrad_amd/csrc/lbl_kernels.hip" $what <<'PY'
import sys
p, what = sys.argv[1], sys.argv[2]
s = open(p).read()
anchor = "        const bool any_far = (iF1 - iB) + (iC - iF2) > 0;"
assert anchor in s
if what == "nogauss":
    s = s.replace("const bool gauss = valid && mine && max(0, max(ci - whi, wlo - ci)) < dgi;", "const bool gauss = false && mine;")
    s = s.replace("        const bool gauss = valid && max(0, max(ci - whi, wlo - ci)) < dgi;\n        const unsigned long long gmask = __ballot(gauss);\n        if (gmask) {                                   // rare",
                  "        const bool gauss = false;\n        const unsigned long long gmask = __ballot(gauss);\n        if (gmask) {                                   // rare")
elif what == "nofar":
    s = s.replace(anchor, "        iF1 = iB; { const int t_ = iF2; iF2 = iC; iB = iF1; (void)t_; }\n        iB = iF1;  /* far lines dropped: near range keeps its bounds below */\n" + anchor)
    # keep the near range: restore by recomputing from the table is not possible here, so instead skip the calls
    s = s.replace("        iF1 = iB; { const int t_ = iF2; iF2 = iC; iB = iF1; (void)t_; }\n        iB = iF1;  /* far lines dropped: near range keeps its bounds below */\n", "")
    s = s.replace("        if (any_far) {\n            double C[FF_NT];", "        if (false && any_far) {\n            double C[FF_NT];")
elif what == "noedge":
    s = s.replace(anchor, "        iA = iB; iD = iC;\n" + anchor)
elif what == "nonear":
    line = "accumulate_lines<R, 1>(J.hot, J.cold, iF1, iF2, iB, iC, wlo, whi, x0, Hf, lh, lc, lane, S, G, 64, LS, part);"
    assert s.count(line) == 2
    s = s.replace(line, "/* near lines dropped */;")
open(p, "w").write(s)
PY
  make -C $T/pyrad_amd/csrc -j4 > $T/build.log 2>&1 || { grep -E "error" $T/build.log; exit 1; }
  cp $T/pyrad_amd/lib/libpyrad_hip.so $ROOT/scripts/bin/libpyrad_hip_abl_$what.so
  rm -rf $T
  echo built $what
done
