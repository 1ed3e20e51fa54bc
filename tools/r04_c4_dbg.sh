cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
RTS_DEBUG_COOP=1 python bench.py --config c4 --steps 12 --warmup 12 --no-cpu-baseline > gpurun_out/r04x_c4_dbg.json 2> gpurun_out/r04x_c4_dbg.err
grep -E "end:|tile |records" gpurun_out/r04x_c4_dbg.err | cut -c1-200 | head -120
