#!/bin/bash
# Timing ablations (results are WRONG by construction; timing only) with the diagnostic build of the library
# (scripts/make_diag_lib.sh: -DLBL_DIAG; the production library has no such code):
#   gpurun -- 'bash scripts/make_diag_lib.sh && bash scripts/ablate.sh C3'
# debug_ablate bits: far-field kernel 8 no edge lines, 256 no far-field series, 512 no near lines (+ their Gaussian runs); 1024 no Gaussian part anywhere (K1 writes reach 0);
# skewed-range kernel 1 no Gaussian walk, 2 no partial-cover Lorentz walk, 4 no full-cover Lorentz walk;
# column step 16 memory traffic only, 32 arithmetic only; layer sweep 64 plain instead of streaming loads / stores.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
WL=${1:-C3}
for bits in ${ABLATE_BITS:-0 8 256 512 1024 1 2 4 16 32}; do
  PYRAD_HIP_LIB=$R/scripts/bin/libpyrad_hip_diag.so python3 $R/bench.py --workload $WL --steps 20 --warmup 3 --blocks 1 \
      --no-cpu-baseline --no-api-path --no-direct-pass $ABLATE_ARGS --set debug_ablate=$bits |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('debug_ablate=$bits', 'step %.4f ms' % d['ms_per_step'], {k: round(v, 4) for k, v in d['kernel_ms_per_step'].items() if k != 'source'}, 'ablated' if d.get('ablated') else '')"
done
