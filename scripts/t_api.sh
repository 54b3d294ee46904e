cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_abi.py tests/test_gpu_model.py tests/test_gpu_schedule.py -q -x -m gpu > gpurun_out/t3.log 2>&1 || { tail -40 gpurun_out/t3.log; exit 1; }
tail -2 gpurun_out/t3.log
timeout -k 10 500 python bench.py --workload C5 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/c5.json 2> gpurun_out/c5.err; python -c "
import json; d=json.load(open('gpurun_out/c5.json')); a=d['api_path']; print(d['ms_per_step'], a['ms_per_call'], a['ms_change_pressure'], a['ms_change_pressure_mutator'], a['ms_first_call'])"
timeout -k 10 500 python bench.py --workload C3 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/c3.json 2> gpurun_out/c3.err; python -c "
import json; d=json.load(open('gpurun_out/c3.json')); a=d['api_path']; print(d['ms_per_step'], {k: round(v,3) for k,v in a.items() if isinstance(v,float)})"
