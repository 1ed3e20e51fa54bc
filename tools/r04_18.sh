cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
T=r04q
python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gputests.log 2>&1; tail -3 gpurun_out/${T}_gputests.log
for r in 0 3 6 7; do echo "part $r/8: $(RTS_SHARD=8 RTS_SHARD_PART=$r RTS_VERBOSE=1 python tools/trace_bench.py c4 8 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-300)" >> gpurun_out/${T}_c4_eighths.log; done
echo "whole: $(RTS_VERBOSE=1 python tools/trace_bench.py c4 10 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-330)" >> gpurun_out/${T}_c4_eighths.log
echo "whole, old lib: $(RTS_AMD_LIB=$PWD/variants/librts_r04_before_xcd.so RTS_VERBOSE=1 python tools/trace_bench.py c4 10 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-330)" >> gpurun_out/${T}_c4_eighths.log
cat gpurun_out/${T}_c4_eighths.log
echo "c3: $(python tools/trace_bench.py c3 10 | tail -1)"; echo "c3 old: $(RTS_AMD_LIB=$PWD/variants/librts_r04_before_xcd.so python tools/trace_bench.py c3 10 | tail -1)"
python bench.py --no-cpu-baseline --steps 64 > gpurun_out/${T}_bench_c3.json 2>/dev/null; python tools/bench_line.py gpurun_out/${T}_bench_c3.json
