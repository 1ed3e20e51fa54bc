cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
T=r04s
for r in 0 1 2 3 4 5 6 7; do echo "part $r/8: $(RTS_SHARD=8 RTS_SHARD_PART=$r RTS_VERBOSE=1 python tools/trace_bench.py c4 8 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-300)" >> gpurun_out/${T}_c4_eighths.log; done
echo "whole: $(RTS_VERBOSE=1 python tools/trace_bench.py c4 10 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-330)" >> gpurun_out/${T}_c4_eighths.log
cat gpurun_out/${T}_c4_eighths.log
echo "c3: $(python tools/trace_bench.py c3 10 | tail -1)"
python -m pytest tests -m gpu -x -q -k "cooperative or c4 or fuzz or interleaved" 2>&1 | tail -2
RTS_SHARD=8 RTS_SHARD_PART=6 RTS_TIMELINE_LAUNCHES=5 python tools/timeline.py c4 > gpurun_out/${T}_timeline_c4_part6.log 2>&1; grep -E "stats|sorted tile|balanced|slowest" gpurun_out/${T}_timeline_c4_part6.log | cut -c1-300
