cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
T=r04v
python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gputests.log 2>&1; tail -5 gpurun_out/${T}_gputests.log
python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/${T}_bench_c3_steps20.json 2> gpurun_out/${T}_bench.err; python tools/bench_line.py gpurun_out/${T}_bench_c3_steps20.json; tail -3 gpurun_out/${T}_bench.err
python -c "
import json; j=json.loads(open('gpurun_out/${T}_bench_c3_steps20.json').read().strip().splitlines()[-1]); c=j['config']
print({k:c[k] for k in ('hit_fraction','bounding_sphere_fraction','walked_segments_per_pulse','walked_Mrays_per_s','dense_control_Gseg_per_s','segments_per_pulse')})"
