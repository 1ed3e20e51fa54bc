cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
for i in 1 2 3; do RTS_LAP=1 python bench.py --no-cpu-baseline --config sphere6 --steps 256 > gpurun_out/r04f_bench_sphere6_$i.json 2> gpurun_out/r04f_bench_sphere6_$i.err; python tools/bench_line.py gpurun_out/r04f_bench_sphere6_$i.json; grep "rts lap" gpurun_out/r04f_bench_sphere6_$i.err | sed -n 3,4p; done
