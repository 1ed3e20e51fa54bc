cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
T=r04k
python -m pytest tests -m gpu -x -q -k "cooperative" > gpurun_out/${T}_newtests.log 2>&1; tail -3 gpurun_out/${T}_newtests.log
for P in 1 2 4 8 1 2 4 8; do echo "RTS_COOP_SPREAD=$P: $(RTS_COOP_SPREAD=$P RTS_VERBOSE=1 python tools/trace_bench.py c4 10 2>&1 | tail -2 | tr '\n' ' ')" >> gpurun_out/${T}_c4_spread.log; done
cat gpurun_out/${T}_c4_spread.log | cut -c1-330
