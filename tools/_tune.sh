cd $GRAFT_REPO_ROOT
for b in 1 2 3 4 6; do
  echo "budget $b: $(RTS_SPLIT_BUDGET=$b python3 tools/count_stats.py 2>/dev/null | grep -E '^c3 |^c3narrow ' | tr '\n' '|')" >> gpurun_out/tune.log
  echo "   c3 $(RTS_SPLIT_BUDGET=$b python3 tools/trace_bench.py c3 12 | tail -1 | cut -c30-140)" >> gpurun_out/tune.log
  echo "   dense $(RTS_SPLIT_BUDGET=$b python3 tools/trace_bench.py c3narrow 8 | tail -1 | cut -c30-140)" >> gpurun_out/tune.log
done
cat gpurun_out/tune.log
