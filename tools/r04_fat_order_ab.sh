cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
L=gpurun_out/r04_fat_order_ab.log; : > $L
for rep in 1 2; do for lib in variants/librts_before.so rts_amd/librts_amd.so; do
  for cfg in "c4 --inflight 1 --steps 12 --warmup 6" "c4 --steps 24 --warmup 12" "c3 --steps 128"; do
    RTS_AMD_LIB=$lib python bench.py --no-cpu-baseline --config $cfg > gpurun_out/r04x_fo.json 2>/dev/null
    echo "${lib:-tree} --config $cfg: $(python tools/bench_line.py gpurun_out/r04x_fo.json | cut -c15-95)" | tee -a $L
  done; done; done
