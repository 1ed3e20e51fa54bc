# lane statistics of the walk under the asynchronous-bounce schedule (counting build) + C4 with and without the cooperative kernel
cd "${GRAFT_REPO_ROOT:?}"
T=${1:-async_stats}
for s in 0:8:7500 64:64:7500 48:8:7500 32:8:7500 16:8:7500 8:8:0 2:2:0; do
  i0=${s%%:*}; r=${s#*:}; i1=${r%%:*}; ag=${r#*:}
  echo "idle0 $i0 idle1 $i1 age $ag: $(RTS_ASYNC_IDLE0=$i0 RTS_ASYNC_IDLE1=$i1 RTS_ASYNC_AGE=$ag timeout -k 10 300 python3 tools/count_stats.py c3 c3narrow | tr '\n' '#')" >> gpurun_out/${T}.log
done
for f in 0.5 0; do
  for s in 0:8:7500 64:8:7500 64:2:7500 64:8:1875; do
    i0=${s%%:*}; r=${s#*:}; i1=${r%%:*}; ag=${r#*:}
    echo "coop_frac $f idle0 $i0 idle1 $i1 age $ag c4: $(RTS_COOP_FRAC=$f RTS_ASYNC_IDLE0=$i0 RTS_ASYNC_IDLE1=$i1 RTS_ASYNC_AGE=$ag timeout -k 10 300 python3 tools/trace_bench.py c4 8 | tail -1)" >> gpurun_out/${T}.log
  done
done
cat gpurun_out/${T}.log
