cd $GRAFT_REPO_ROOT
T=${1:-ab}
for w in c3 c3narrow c3empty c3; do
  echo "before $w: $(RTS_AMD_LIB=variants/librts_before.so python3 tools/trace_bench.py $w 12 | tail -1)" >> gpurun_out/${T}.log
  echo "after  $w: $(python3 tools/trace_bench.py $w 12 | tail -1)" >> gpurun_out/${T}.log
  echo "nocoop $w: $(RTS_COOP_AFTER=0 python3 tools/trace_bench.py $w 12 | tail -1)" >> gpurun_out/${T}.log
done
cat gpurun_out/${T}.log
