cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
T=r04p
echo "old lib part 6: $(RTS_AMD_LIB=$PWD/variants/librts_r04_before_xcd.so RTS_SHARD=8 RTS_SHARD_PART=6 RTS_VERBOSE=1 python tools/trace_bench.py c4 8 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-300)"
echo "r03b lib part 6: $(RTS_AMD_LIB=$PWD/variants/librts_r03b.so RTS_SHARD=8 RTS_SHARD_PART=6 RTS_VERBOSE=1 python tools/trace_bench.py c4 8 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-300)"
echo "new, order in four launches: $(RTS_ORDER_FUSED=0 RTS_SHARD=8 RTS_SHARD_PART=6 RTS_VERBOSE=1 python tools/trace_bench.py c4 8 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-300)"
echo "new, debug: $(RTS_DEBUG_COOP=1 RTS_SHARD=8 RTS_SHARD_PART=6 RTS_VERBOSE=1 python tools/trace_bench.py c4 4 2>&1 | tail -8 | cut -c1-200)"
