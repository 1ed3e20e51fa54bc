# the receiver WINDOW screen of the pre-filter (RTS_RX_WINDOW_SCREEN=0: the pre-filter only asks whether a capture sphere is reached) and the dead-tile
# batches behind it, on BASELINE configs[3] (every ray of its beam crosses a capture sphere) and configs[2]: same-box A/B.  usage: tools/screen_ab.sh <tag>
cd "${GRAFT_REPO_ROOT:?}"
T=${1:-screen_ab}; : > gpurun_out/${T}.log
for w in c4 c3 c4; do
  echo "no screen, per tile  $w: $(RTS_RX_WINDOW_SCREEN=0 RTS_DEAD_BATCH=0 python3 tools/trace_bench.py $w 10 | tail -1)" >> gpurun_out/${T}.log
  echo "screen, per tile     $w: $(RTS_DEAD_BATCH=0 python3 tools/trace_bench.py $w 10 | tail -1)" >> gpurun_out/${T}.log
  echo "screen + batches     $w: $(python3 tools/trace_bench.py $w 10 | tail -1)" >> gpurun_out/${T}.log
done
RTS_RX_WINDOW_SCREEN=0 RTS_DEAD_BATCH=0 python3 bench.py --no-cpu-baseline --config c4 --steps 24 --warmup 12 > gpurun_out/${T}_bench_c4_off.json 2>/dev/null
python3 bench.py --no-cpu-baseline --config c4 --steps 24 --warmup 12 > gpurun_out/${T}_bench_c4_on.json 2>/dev/null
python3 - <<PY >> gpurun_out/${T}.log
import json
for n in ("c4_off","c4_on"):
    d=json.loads(open("gpurun_out/${T}_bench_%s.json"%n).read().strip().splitlines()[-1]); print("bench",n,round(d["value"]),"Mrays/s",round(d["ms_per_step"],4),"ms/pulse; serial kernel",round(d["roofline"]["kernel_ms_serial"],4))
PY
cat gpurun_out/${T}.log
