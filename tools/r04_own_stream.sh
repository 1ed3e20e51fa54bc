cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
L=gpurun_out/r04_trace_own_stream_ab.log; : > $L
for rep in 1 2; do for v in 1 0; do
  for cfg in "c3 --inflight 1 --steps 64" "sphere6 --inflight 1 --steps 64" "c2 --inflight 1 --steps 64" "c4 --inflight 1 --steps 12 --warmup 6" "c3 --steps 64"; do
    RTS_TRACE_OWN_STREAM=$v python bench.py --no-cpu-baseline --config $cfg > gpurun_out/r04x_os.json 2>/dev/null
    echo "RTS_TRACE_OWN_STREAM=$v --config $cfg: $(python tools/bench_line.py gpurun_out/r04x_os.json | cut -c15-95)" | tee -a $L
  done; done; done
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
