cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
T=r04c
python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gputests.log 2>&1; tail -3 gpurun_out/${T}_gputests.log
RTS_LAP=1 python bench.py --no-cpu-baseline --config sphere6 --steps 256 > gpurun_out/${T}_bench_sphere6.json 2> gpurun_out/${T}_bench_sphere6.err; python tools/bench_line.py gpurun_out/${T}_bench_sphere6.json; grep "rts lap" gpurun_out/${T}_bench_sphere6.err | head -4
RTS_LAP=1 python bench.py --no-cpu-baseline --steps 256 > gpurun_out/${T}_bench_c3.json 2> gpurun_out/${T}_bench_c3.err; python tools/bench_line.py gpurun_out/${T}_bench_c3.json; grep "rts lap" gpurun_out/${T}_bench_c3.err | head -4
RTS_LAP=1 tools/adapter_bench_bin 216 256 3 6 6 1 d > gpurun_out/${T}_adapter_lap.json 2> gpurun_out/${T}_adapter_lap.err; grep "rts lap" gpurun_out/${T}_adapter_lap.err | tail -4
