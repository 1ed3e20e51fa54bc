cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
T=r04o
for r in 0 1 2 3 4 5 6 7; do echo "part $r/8: $(RTS_SHARD=8 RTS_SHARD_PART=$r RTS_VERBOSE=1 python tools/trace_bench.py c4 8 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-330)" >> gpurun_out/${T}_c4_eighths.log; done
echo "whole: $(RTS_VERBOSE=1 python tools/trace_bench.py c4 8 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-330)" >> gpurun_out/${T}_c4_eighths.log
cat gpurun_out/${T}_c4_eighths.log
