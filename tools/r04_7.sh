cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
python bench.py --no-cpu-baseline --config sphere6 --steps 256 --no-bind > gpurun_out/r04g_1.json 2>/dev/null; python tools/bench_line.py gpurun_out/r04g_1.json
python bench.py --no-cpu-baseline --config sphere6 --steps 256 > gpurun_out/r04g_2.json 2>/dev/null; python tools/bench_line.py gpurun_out/r04g_2.json
python bench.py --no-cpu-baseline --config sphere6 --steps 256 --fused-post > gpurun_out/r04g_3.json 2>/dev/null; python tools/bench_line.py gpurun_out/r04g_3.json
python bench.py --no-cpu-baseline --config c2 --steps 256 > gpurun_out/r04g_4.json 2>/dev/null; python tools/bench_line.py gpurun_out/r04g_4.json
nproc; lscpu | grep -E "Model name|Socket|NUMA node" | head -6
