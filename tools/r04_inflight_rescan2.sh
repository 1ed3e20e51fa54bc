cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out; L=gpurun_out/r04_inflight_rescan.log
for q in 8 16; do for n in 3 4 5; do
  GPU_MAX_HW_QUEUES=$q python bench.py --no-cpu-baseline --steps 128 --inflight $n > gpurun_out/r04x_if.json 2>/dev/null
  echo "GPU_MAX_HW_QUEUES=$q --inflight $n: $(python tools/bench_line.py gpurun_out/r04x_if.json | cut -c15-75)" | tee -a $L
done; done
