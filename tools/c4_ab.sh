cd "${GRAFT_REPO_ROOT:?}"
T=${1:-c4_ab}
for rep in 1 2 3; do
  echo "r03b    c4: $(RTS_AMD_LIB=variants/librts_r03b.so python3 tools/trace_bench.py c4 10 | tail -1)" >> gpurun_out/${T}.log
  echo "now     c4: $(python3 tools/trace_bench.py c4 10 | tail -1)" >> gpurun_out/${T}.log
  echo "now, radix order c4: $(RTS_TILE_SORT=radix python3 tools/trace_bench.py c4 10 | tail -1)" >> gpurun_out/${T}.log
  echo "now, no coop c4: $(RTS_COOP_FRAC=0 python3 tools/trace_bench.py c4 6 | tail -1)" >> gpurun_out/${T}.log
  echo "r03b, no coop c4: $(RTS_COOP_FRAC=0 RTS_AMD_LIB=variants/librts_r03b.so python3 tools/trace_bench.py c4 6 | tail -1)" >> gpurun_out/${T}.log
done
cat gpurun_out/${T}.log
