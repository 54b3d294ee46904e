R=$GRAFT_REPO_ROOT
for l in diag430 diag338 diag430 diag338; do
  export PYRAD_HIP_LIB=$R/scripts/bin/libpyrad_hip_$l.so
  for ab in 0 1024; do
    python3 $R/bench.py --steps 30 --warmup 3 --blocks 1 --no-cpu-baseline --no-api-path --no-direct-pass --legs none --workload C3 --step merged --set debug_ablate=$ab 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$l', 'ablate $ab', 'step %.4f' % d['ms_per_step'], 'K2 %.4f' % d['kernel_ms_per_step']['xsec_accumulate'])"
  done
done
