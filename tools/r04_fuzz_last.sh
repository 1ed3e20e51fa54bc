cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
timeout -k 10 900 python tools/fuzz_equal.py 1500 8500000 --oracle > gpurun_out/r04_fuzz5_standard.log 2>&1; tail -2 gpurun_out/r04_fuzz5_standard.log | cut -c1-250
RTS_COOP_FRAC=1e-12 RTS_COOP_FLOOR=0 RTS_COOP_SEG=0 timeout -k 10 600 python tools/fuzz_equal.py 300 8600000 --big --oracle > gpurun_out/r04_fuzz5_big_forced_coop.log 2>&1; tail -2 gpurun_out/r04_fuzz5_big_forced_coop.log | cut -c1-250
timeout -k 10 300 python tools/fuzz_aggregate.py 600 8700000 > gpurun_out/r04_fuzz5_aggregate.log 2>&1; tail -1 gpurun_out/r04_fuzz5_aggregate.log | cut -c1-250
