cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
T=r04j
python -m pytest tests -m gpu -x -q -k "affine or cooperative or c4 or fuzz" > gpurun_out/${T}_newtests.log 2>&1; tail -5 gpurun_out/${T}_newtests.log
for i in 1 2; do echo "coop lists by XCD: $(RTS_VERBOSE=1 python tools/trace_bench.py c4 10 2>&1 | tail -2 | tr '\n' ' ')" >> gpurun_out/${T}_c4.log; done
cat gpurun_out/${T}_c4.log
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for p in tcc fetch; do
    case $p in tcc) C="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum";; fetch) C="FETCH_SIZE";; esac
    OUT=$ROOT/gpurun_out/pmc_r04j_c4/$p; mkdir -p $ROOT/gpurun_out/pmc_r04j_c4
    timeout -k 10 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/trace_bench.py c4 6 > $OUT.log 2>&1
    find $OUT -name "*_kernel_trace.csv" -delete; find $OUT -name "*_agent_info.csv" -delete
    for f in $(find $OUT -name "*_counter_collection.csv"); do { head -1 $f; grep k_trace $f; } > $f.tmp && mv $f.tmp $f; done
done
python3 $ROOT/tools/pmc_quick.py r04j_c4 tcc fetch
