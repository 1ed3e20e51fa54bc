cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
L=gpurun_out/r04_c3_coop_big_scan.log; : > $L
for big in 0 0.3 0.5 0.7 1.0 1.5; do
  echo "RTS_COOP_BIG=$big: $(RTS_COOP_BIG=$big RTS_VERBOSE=1 python tools/trace_bench.py c3 24 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-400)" | tee -a $L
done
