for v in 0 1 2; do for r in 1 2 4 8; do
 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --variant $v --points-per-lane $r | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C2 v=$v R=$r', 'acc_ms=%.3f'%d['kernel_ms_per_step']['xsec_accumulate'], 'evals/s=%.3e'%d['valu_f64']['kernel_evals_per_s'])"
done; done
for r in 2 4 8; do
 python bench.py --workload C3 --steps 3 --warmup 1 --no-cpu-baseline --variant 2 --points-per-lane $r | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('C3 v=2 R=$r', 'acc_ms=%.3f'%d['kernel_ms_per_step']['xsec_accumulate'], 'evals/s=%.3e'%d['valu_f64']['kernel_evals_per_s'], d['kernel_ms_per_step'])"
done
