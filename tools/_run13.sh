cd $GRAFT_REPO_ROOT
python tools/cpu_baseline.py > gpurun_out/r02c_cpu_baseline.log 2>&1; cat gpurun_out/r02c_cpu_baseline.log
python bench.py --steps 20 --warmup 5 > gpurun_out/r02c_bench_c3_steps20.json 2> gpurun_out/r02c_bench_c3_steps20.err; python -c "
import json
j=json.loads(open('gpurun_out/r02c_bench_c3_steps20.json').read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'], j['cpu_baseline'])"
