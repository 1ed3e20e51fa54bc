cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
L=gpurun_out/r04_sched_strategy_ab.log; : > $L
for rep in 1 2; do for v in rts_amd/librts_amd.so variants/librts_max-ilp.so variants/librts_max-memory-clause.so variants/librts_iterative-minreg.so variants/librts_iterative-ilp.so; do
  echo "$v c3narrow: $(RTS_AMD_LIB=$v python3 tools/trace_bench.py c3narrow 10 | tail -1 | cut -c30-140)" | tee -a $L
  RTS_AMD_LIB=$v python bench.py --no-cpu-baseline --steps 128 > gpurun_out/r04x_ss.json 2>/dev/null; echo "$v bench c3 128 steps: $(python tools/bench_line.py gpurun_out/r04x_ss.json | cut -c15-75)" | tee -a $L
done; done
