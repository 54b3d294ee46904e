run() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$*', 'acc_ms=%.4f'%d['kernel_ms_per_step']['xsec_accumulate'], 'step_ms=%.4f'%d['ms_per_step'], '%.3e'%d['value'])"; }
for r in 2 4; do for ls in 1 2 4; do run --workload C2 --points-per-lane $r --line-split $ls; done; done
run --workload C2 --variant 4 --points-per-lane 4
for r in 4; do for ls in 1 2 4; do run --workload C3 --points-per-lane $r --line-split $ls; done; done
run --workload C3 --variant 4 --points-per-lane 4
