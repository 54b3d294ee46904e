run() { python bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', 'acc_ms=%.3f'%d['kernel_ms_per_step']['xsec_accumulate'], 'evals/s=%.3e'%d['valu_f64']['kernel_evals_per_s'], 'step_ms=%.3f'%d['ms_per_step'])"; }
for o in 0 1; do for r in 2 4 8; do for ls in 2 4; do run --workload C2 --tile-order $o --points-per-lane $r --line-split $ls; done; done; done
for o in 0 1; do for r in 4 8; do run --workload C3 --tile-order $o --points-per-lane $r --line-split 1; done; done
