cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
T=r04aa
python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gputests.log 2>&1; tail -3 gpurun_out/${T}_gputests.log
python bench.py --no-cpu-baseline --config c5 --steps 512 --warmup 16 > gpurun_out/${T}_c5.json 2>/dev/null; echo "c5: $(python tools/bench_line.py gpurun_out/${T}_c5.json | cut -c1-110)"
for r in 0 1 2 3 4 5 6 7; do echo "part $r/8: $(RTS_SHARD=8 RTS_SHARD_PART=$r RTS_VERBOSE=1 python tools/trace_bench.py c4 8 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-300)" >> gpurun_out/${T}_c4_eighths.log; done
echo "whole: $(RTS_VERBOSE=1 python tools/trace_bench.py c4 10 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-330)" >> gpurun_out/${T}_c4_eighths.log
cut -c1-120 gpurun_out/${T}_c4_eighths.log
python bench.py --no-cpu-baseline --steps 128 > gpurun_out/${T}_c3.json 2>/dev/null; echo "c3: $(python tools/bench_line.py gpurun_out/${T}_c3.json | cut -c1-110)"
