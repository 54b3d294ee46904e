cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_abi.py tests/test_gpu_model.py tests/test_gpu_merged.py tests/test_gpu_schedule.py -q -x -m gpu > gpurun_out/t2.log 2>&1 || { tail -40 gpurun_out/t2.log; exit 1; }
tail -2 gpurun_out/t2.log
timeout -k 10 300 python scripts/profile_api_column.py > gpurun_out/api_prof.log 2>&1; head -4 gpurun_out/api_prof.log
