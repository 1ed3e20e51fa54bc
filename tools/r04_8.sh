cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
RTS_LAP=1 RTS_PY_LAP=1 python bench.py --no-cpu-baseline --config c2 --steps 256 > gpurun_out/r04h_1.json 2> gpurun_out/r04h_1.err; python tools/bench_line.py gpurun_out/r04h_1.json; grep -E "python side|rts lap" gpurun_out/r04h_1.err | tail -4
RTS_LAP=1 RTS_PY_LAP=1 python bench.py --no-cpu-baseline --config c2 --steps 64 > gpurun_out/r04h_2.json 2> gpurun_out/r04h_2.err; python tools/bench_line.py gpurun_out/r04h_2.json; grep -E "python side|rts lap" gpurun_out/r04h_2.err | tail -4
