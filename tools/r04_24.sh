cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
T=r04w
python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gputests.log 2>&1; tail -3 gpurun_out/${T}_gputests.log
timeout -k 10 400 python tools/fuzz_equal.py 1000 410000 --oracle > gpurun_out/${T}_fuzz_standard.log 2>&1; tail -2 gpurun_out/${T}_fuzz_standard.log | cut -c1-300
RTS_COOP_FRAC=1e-12 RTS_COOP_FLOOR=0 RTS_COOP_SEG=0 timeout -k 10 300 python tools/fuzz_equal.py 200 420000 --big --oracle > gpurun_out/${T}_fuzz_big_forced_coop.log 2>&1; tail -2 gpurun_out/${T}_fuzz_big_forced_coop.log | cut -c1-300
RTS_XCD_AFFINE=1 RTS_GRID_MULT=1 timeout -k 10 300 python tools/fuzz_equal.py 200 430000 --big --oracle > gpurun_out/${T}_fuzz_big_xcd_affine.log 2>&1; tail -2 gpurun_out/${T}_fuzz_big_xcd_affine.log | cut -c1-300
timeout -k 10 200 python tools/fuzz_aggregate.py 600 440000 > gpurun_out/${T}_fuzz_aggregate.log 2>&1; tail -2 gpurun_out/${T}_fuzz_aggregate.log | cut -c1-300
