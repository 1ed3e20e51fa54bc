cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
timeout -k 10 600 python tools/fuzz_equal.py 700 8800000 --oracle > gpurun_out/r04_fuzz6_standard.log 2>&1; tail -2 gpurun_out/r04_fuzz6_standard.log | cut -c1-250
RTS_COOP_FRAC=1e-12 RTS_COOP_FLOOR=0 RTS_COOP_SEG=0 timeout -k 10 600 python tools/fuzz_equal.py 250 8900000 --big --oracle > gpurun_out/r04_fuzz6_big_forced_coop.log 2>&1; tail -2 gpurun_out/r04_fuzz6_big_forced_coop.log | cut -c1-250
