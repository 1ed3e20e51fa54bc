cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
L=gpurun_out/r04_fresh_processes_gc.log
echo "# python bench.py --no-cpu-baseline (256 steps, 3 in flight), fresh processes in turn: collector off in the timed interval (the default) / left on (RTS_BENCH_GC=1)" > $L
for i in 1 2 3 4 5 6 7 8; do
  python bench.py --no-cpu-baseline > gpurun_out/r04x_f.json 2>/dev/null; echo "gc off  #$i: $(python tools/bench_line.py gpurun_out/r04x_f.json | cut -c1-175)" | tee -a $L
  RTS_BENCH_GC=1 python bench.py --no-cpu-baseline > gpurun_out/r04x_f.json 2>/dev/null; echo "gc ON   #$i: $(python tools/bench_line.py gpurun_out/r04x_f.json | cut -c1-175)" | tee -a $L
done
