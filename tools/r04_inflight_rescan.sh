cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out; L=gpurun_out/r04_inflight_rescan.log; : > $L
for rep in 1 2; do for n in 2 3 4 5; do
  python bench.py --no-cpu-baseline --steps 128 --inflight $n > gpurun_out/r04x_if.json 2>/dev/null
  echo "--inflight $n: $(python tools/bench_line.py gpurun_out/r04x_if.json | cut -c15-75)" | tee -a $L
done; done
