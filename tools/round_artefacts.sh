cd "${GRAFT_REPO_ROOT:?}"
python bench.py --steps 20 --warmup 5 > gpurun_out/r02d_bench_c3_steps20.json 2> gpurun_out/r02d_bench_c3_steps20.err; tail -c 1500 gpurun_out/r02d_bench_c3_steps20.json; echo
python bench.py --no-cpu-baseline > gpurun_out/r02d_bench_c3_default.json 2>&1
python bench.py --no-cpu-baseline --config c3ecef --steps 64 > gpurun_out/r02d_bench_c3ecef.json 2>&1
python bench.py --no-cpu-baseline --config c2 --steps 64 > gpurun_out/r02d_bench_c2.json 2>&1
python bench.py --no-cpu-baseline --config c2file --steps 64 > gpurun_out/r02d_bench_c2file.json 2>&1
for f in c3_default c3ecef c2 c2file; do python -c "
import json,sys
j=json.loads(open('gpurun_out/r02d_bench_$f.json').read().strip().splitlines()[-1])
r=j['roofline']
print('$f', round(j['value']), 'Mrays/s', round(j['ms_per_step'],3), 'ms/pulse | serial', round(r['kernel_ms_serial'],3), 'hit', round(r['hit_fraction'],3), 'V,T', round(r['nodes_per_segment'],2), round(r['tri_tests_per_segment'],2), '| bound', r['bound'], r['frac'], 'valu', r.get('valu_issue_frac'), 'td', r.get('vmem_return_path_frac'))"; done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02d -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r02d_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/r02d_bench_under_rocprof.err; cd $GRAFT_REPO_ROOT
ls gpurun_out/prof_r02d/*/ | head
python tools/cpu_baseline.py > gpurun_out/r02d_cpu_baseline.log 2>&1; cat gpurun_out/r02d_cpu_baseline.log
