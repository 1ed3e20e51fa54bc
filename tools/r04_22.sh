cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
T=r04u
for st in 200 100; do
for r in 3 6; do echo "steps $st part $r/8: $(RTS_COOP_STEPS=$st RTS_SHARD=8 RTS_SHARD_PART=$r RTS_VERBOSE=1 python tools/trace_bench.py c4 7 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-110)" >> gpurun_out/${T}_c4_steps.log; done
echo "steps $st whole: $(RTS_COOP_STEPS=$st RTS_VERBOSE=1 python tools/trace_bench.py c4 8 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-130)" >> gpurun_out/${T}_c4_steps.log
echo "steps $st c3: $(RTS_COOP_STEPS=$st RTS_VERBOSE=1 python tools/trace_bench.py c3 10 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-200)" >> gpurun_out/${T}_c4_steps.log
done
cat gpurun_out/${T}_c4_steps.log
