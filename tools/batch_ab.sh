# dead-tile batches of the trace kernel (RTS_DEAD_BATCH=0: every tile by a whole wave): same-box A/B.  usage: tools/batch_ab.sh <tag>
cd "${GRAFT_REPO_ROOT:?}"
T=${1:-batch_ab}; : > gpurun_out/${T}.log
for w in c3 c3empty c2 c4 c3 c3narrow; do
  echo "per tile $w: $(RTS_DEAD_BATCH=0 python3 tools/trace_bench.py $w 12 | tail -1)" >> gpurun_out/${T}.log
  echo "batches  $w: $(python3 tools/trace_bench.py $w 12 | tail -1)" >> gpurun_out/${T}.log
done
RTS_DEAD_BATCH=0 python3 bench.py --no-cpu-baseline --steps 64 > gpurun_out/${T}_bench_off.json 2>/dev/null
python3 bench.py --no-cpu-baseline --steps 64 > gpurun_out/${T}_bench_on.json 2>/dev/null
RTS_DEAD_BATCH=0 python3 bench.py --no-cpu-baseline --config c4 --steps 24 --warmup 12 > gpurun_out/${T}_bench_c4_off.json 2>/dev/null
python3 bench.py --no-cpu-baseline --config c4 --steps 24 --warmup 12 > gpurun_out/${T}_bench_c4_on.json 2>/dev/null
python3 - <<PY >> gpurun_out/${T}.log
import json
for n in ("off","on","c4_off","c4_on"):
    d=json.loads(open("gpurun_out/${T}_bench_%s.json"%n).read().strip().splitlines()[-1]); print("bench",n,round(d["value"]),"Mrays/s",round(d["ms_per_step"],4),"ms/pulse; serial kernel",round(d["roofline"]["kernel_ms_serial"],4))
PY
cat gpurun_out/${T}.log
