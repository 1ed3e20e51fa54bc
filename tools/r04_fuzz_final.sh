cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
RTS_COOP_FRAC=1e-12 RTS_COOP_FLOOR=0 RTS_COOP_SEG=0 timeout -k 10 600 python tools/fuzz_equal.py 200 8200000 --big --oracle > gpurun_out/r04_fuzz2_big_forced_coop.log 2>&1; tail -2 gpurun_out/r04_fuzz2_big_forced_coop.log | cut -c1-250
