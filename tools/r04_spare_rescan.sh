cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out; L=gpurun_out/r04_grid_spare_rescan.log; : > $L
for rep in 1 2; do for sp in default 0 64 160 256 384; do
  if [ $sp = default ]; then python bench.py --no-cpu-baseline --steps 128 > gpurun_out/r04x_sp.json 2>/dev/null; else RTS_GRID_SPARE=$sp python bench.py --no-cpu-baseline --steps 128 > gpurun_out/r04x_sp.json 2>/dev/null; fi
  echo "RTS_GRID_SPARE=$sp: $(python tools/bench_line.py gpurun_out/r04x_sp.json | cut -c15-75)" | tee -a $L
done; done
