cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
L=gpurun_out/r04_coop_three_waves_ab.log; : > $L
for rep in 1 2; do for lib in variants/librts_before.so rts_amd/librts_amd.so; do
  echo "$lib trace_bench c4: $(RTS_AMD_LIB=$lib RTS_VERBOSE=1 python3 tools/trace_bench.py c4 10 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-260)" | tee -a $L
  RTS_AMD_LIB=$lib python bench.py --no-cpu-baseline --config c4 --steps 24 --warmup 12 > gpurun_out/r04x_co.json 2>/dev/null; echo "$lib bench c4: $(python tools/bench_line.py gpurun_out/r04x_co.json | cut -c15-80)" | tee -a $L
  RTS_AMD_LIB=$lib python bench.py --no-cpu-baseline --config c4 --inflight 1 --steps 12 --warmup 6 > gpurun_out/r04x_co.json 2>/dev/null; echo "$lib bench c4 --inflight 1: $(python tools/bench_line.py gpurun_out/r04x_co.json | cut -c15-80)" | tee -a $L
done; done
for lib in variants/librts_before.so rts_amd/librts_amd.so; do RTS_AMD_LIB=$lib python tools/deal_bench.py c4 8 4096 2>&1 | tail -2 | cut -c1-200 | sed "s|^|$lib |" | tee -a $L; done
