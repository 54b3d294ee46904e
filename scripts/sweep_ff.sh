#!/bin/bash
# far-field variant sweep: library build (threshold/terms) x points per lane x line split
run() { lib=$1; shift; PYRAD_HIP_LIB=$lib python bench.py --steps 30 --warmup 3 --no-cpu-baseline --variant 5 "$@" 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$(basename $lib)', '$*', 'acc_ms=%.4f'%d['kernel_ms_per_step']['xsec_accumulate'], 'step_ms=%.4f'%d['ms_per_step'], '%.3e'%d['value'])"; }
D=$PWD/pyrad_amd/lib/libpyrad_hip.so
for lib in $D $PWD/scripts/bin/libpyrad_hip_ff*.so; do
  for ls in 1 2 4; do run $lib --workload C2 --points-per-lane 4 --line-split $ls; done
  for ls in 1 2; do run $lib --workload C3 --points-per-lane 4 --line-split $ls; done
done
for r in 2 8; do for ls in 1 2 4; do run $D --workload C2 --points-per-lane $r --line-split $ls; done; done
for r in 2 8; do run $D --workload C3 --points-per-lane $r --line-split 1; done
