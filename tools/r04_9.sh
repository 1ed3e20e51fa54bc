cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
T=r04i
python -m pytest tests -m gpu -x -q -k "affine or host_mirror or cooperative or ragged or interleaved" > gpurun_out/${T}_newtests.log 2>&1; tail -5 gpurun_out/${T}_newtests.log
for m in 0 auto 0 auto; do echo "RTS_XCD_AFFINE=$m: $(RTS_XCD_AFFINE=$m RTS_VERBOSE=1 python tools/trace_bench.py c4 10 2>&1 | tail -2 | tr '\n' ' ')" >> gpurun_out/${T}_c4_affine_ab.log; done
cat gpurun_out/${T}_c4_affine_ab.log
echo "c3 0: $(RTS_XCD_AFFINE=0 python tools/trace_bench.py c3 10 | tail -1)"; echo "c3 1: $(RTS_XCD_AFFINE=1 python tools/trace_bench.py c3 10 | tail -1)"
