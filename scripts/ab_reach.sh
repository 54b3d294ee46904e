cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_whole_spectrum.py tests/test_gpu_merged.py tests/test_gpu_parity.py -q -x -m gpu > gpurun_out/t5.log 2>&1 || { tail -40 gpurun_out/t5.log; exit 1; }
tail -2 gpurun_out/t5.log
bash scripts/bitcmp_libs.sh base prod
bash scripts/ab_libs.sh "base prod" --workload C3
bash scripts/ab_libs.sh "base prod" --workload C3 --step per-list
bash scripts/ab_libs.sh "base prod" --workload C3 --shard-of 8,4
bash scripts/ab_libs.sh "base prod" --workload C5 | cut -c1-200
bash scripts/ab_libs.sh "base prod" --workload C2
