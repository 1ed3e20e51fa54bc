cd $GRAFT_REPO_ROOT
python -m pytest tests/ -x -q -m gpu > gpurun_out/r02_tests12.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r02_tests12.log
for rep in 1 2; do
echo before; RTS_AMD_LIB=$PWD/variants/librts_before.so python tools/trace_bench.py c3 12; RTS_AMD_LIB=$PWD/variants/librts_before.so python tools/trace_bench.py c3narrow 5; RTS_AMD_LIB=$PWD/variants/librts_before.so python tools/trace_bench.py c3nomesh 8
echo after; python tools/trace_bench.py c3 12; python tools/trace_bench.py c3narrow 5; python tools/trace_bench.py c3nomesh 8
done
python bench.py --steps 64 --warmup 8 --no-cpu-baseline > gpurun_out/r02e_bench_64.json 2>&1; python -c "
import json
j=json.loads(open('gpurun_out/r02e_bench_64.json').read().strip().splitlines()[-1]); print('bench after', j['value'], j['ms_per_step'], j['roofline']['kernel_ms_serial'])"
RTS_AMD_LIB=$PWD/variants/librts_before.so python bench.py --steps 64 --warmup 8 --no-cpu-baseline > gpurun_out/r02e_bench_64_before.json 2>&1; python -c "
import json
j=json.loads(open('gpurun_out/r02e_bench_64_before.json').read().strip().splitlines()[-1]); print('bench before', j['value'], j['ms_per_step'], j['roofline']['kernel_ms_serial'])"
