# A/B sweep of the accumulate kernel's launch shape on the GPU box (prints one line per run)
run() { python bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', 'acc_ms=%.3f'%d['kernel_ms_per_step']['xsec_accumulate'], 'evals/s=%.3e'%d['valu_f64']['kernel_evals_per_s'], 'step_ms=%.3f'%d['ms_per_step'])"; }
run --workload C2
for r in 2 4 8; do for ls in 1 2 4; do run --workload C2 --variant 3 --points-per-lane $r --line-split $ls; done; done
run --workload C3
for r in 4 8; do for ls in 1 2; do run --workload C3 --variant 3 --points-per-lane $r --line-split $ls; done; done
