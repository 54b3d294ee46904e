cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_schedule.py tests/test_gpu_merged.py -q -x -m gpu > gpurun_out/t3.log 2>&1 || { tail -40 gpurun_out/t3.log; exit 1; }
tail -2 gpurun_out/t3.log
bash scripts/pack_time.sh c5 --workload C5
bash scripts/pack_time.sh c3s8 --workload C3 --shard-of 8,2
