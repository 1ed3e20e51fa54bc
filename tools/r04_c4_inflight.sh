cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
python bench.py --no-cpu-baseline --config c5 --steps 1024 --warmup 16 > gpurun_out/r04_bench_c5.json 2>/dev/null && python tools/bench_line.py gpurun_out/r04_bench_c5.json
for n in 1 2 3; do
  python bench.py --no-cpu-baseline --config c4 --steps 24 --warmup 12 --inflight $n > gpurun_out/r04x_c4_inflight$n.json 2>/dev/null && echo "inflight $n: $(python tools/bench_line.py gpurun_out/r04x_c4_inflight$n.json)"
done
