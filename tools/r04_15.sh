cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
T=r04n
for i in 1 2; do
for pr in 3 0; do RTS_POST_PRIO=$pr python bench.py --no-cpu-baseline --steps 64 > gpurun_out/${T}_a.json 2>/dev/null; echo "one prio $pr: $(python tools/bench_line.py gpurun_out/${T}_a.json | cut -c1-200)"; done
RTS_POST_ONE=0 python bench.py --no-cpu-baseline --steps 64 > gpurun_out/${T}_a.json 2>/dev/null; echo "seven: $(python tools/bench_line.py gpurun_out/${T}_a.json | cut -c1-200)"
python bench.py --no-cpu-baseline --steps 64 --four-calls > gpurun_out/${T}_a.json 2>/dev/null; echo "four: $(python tools/bench_line.py gpurun_out/${T}_a.json | cut -c1-200)"
done
RTS_POST_PRIO=3 python bench.py --no-cpu-baseline --steps 64 --inflight 1 > gpurun_out/${T}_a.json 2>/dev/null; echo "one prio 3 inflight 1: $(python tools/bench_line.py gpurun_out/${T}_a.json | cut -c1-200)"
python bench.py --no-cpu-baseline --steps 64 --inflight 1 --four-calls > gpurun_out/${T}_a.json 2>/dev/null; echo "four inflight 1: $(python tools/bench_line.py gpurun_out/${T}_a.json | cut -c1-200)"
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$T -- python3 $GRAFT_REPO_ROOT/bench.py --steps 64 --inflight 1 --no-cpu-baseline > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
find gpurun_out/prof_$T -name "*_kernel_trace.csv" -delete; find gpurun_out/prof_$T -name "*_agent_info.csv" -delete
cat $(find gpurun_out/prof_$T -name "*kernel_stats.csv" | head -1) | cut -d, -f1-8 | head -25
