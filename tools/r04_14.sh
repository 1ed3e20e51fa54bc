cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
T=r04m
python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gputests.log 2>&1; tail -4 gpurun_out/${T}_gputests.log
for i in 1 2; do
python bench.py --no-cpu-baseline --steps 64 > gpurun_out/${T}_bench_c3_one_$i.json 2>/dev/null; python tools/bench_line.py gpurun_out/${T}_bench_c3_one_$i.json
RTS_POST_ONE=0 python bench.py --no-cpu-baseline --steps 64 > gpurun_out/${T}_bench_c3_seven_$i.json 2>/dev/null; python tools/bench_line.py gpurun_out/${T}_bench_c3_seven_$i.json
python bench.py --no-cpu-baseline --steps 64 --four-calls > gpurun_out/${T}_bench_c3_four_$i.json 2>/dev/null; python tools/bench_line.py gpurun_out/${T}_bench_c3_four_$i.json
RTS_AMD_LIB=$PWD/variants/librts_r04_before_xcd.so python bench.py --no-cpu-baseline --steps 64 --four-calls > gpurun_out/${T}_bench_c3_old_$i.json 2>/dev/null; python tools/bench_line.py gpurun_out/${T}_bench_c3_old_$i.json
done
python bench.py --no-cpu-baseline --steps 64 --inflight 1 > gpurun_out/${T}_bench_c3_if1.json 2>/dev/null; python tools/bench_line.py gpurun_out/${T}_bench_c3_if1.json
RTS_AMD_LIB=$PWD/variants/librts_r04_before_xcd.so python bench.py --no-cpu-baseline --steps 64 --inflight 1 --four-calls > gpurun_out/${T}_bench_c3_if1_old.json 2>/dev/null; python tools/bench_line.py gpurun_out/${T}_bench_c3_if1_old.json
python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/${T}_bench_c3_steps20.json 2>/dev/null; python tools/bench_line.py gpurun_out/${T}_bench_c3_steps20.json
