# pipelined bench: handles x deferred collection (bench.py --inflight N --post-lag L)
cd "${GRAFT_REPO_ROOT:?}"
T=${1:-lag_ab}
SETS=${2:-"3:0 3:1 4:1 5:1 4:0"}
for rep in 1 2; do
for s in $SETS; do
  n=${s%%:*}; l=${s#*:}
  python3 bench.py --steps 128 --warmup 8 --no-cpu-baseline --inflight $n --post-lag $l > gpurun_out/${T}_${n}_${l}.json 2>/dev/null
  python3 - <<PY >> gpurun_out/${T}.log
import json
d=json.loads(open("gpurun_out/${T}_${n}_${l}.json").read().strip().splitlines()[-1])
print("inflight $n lag $l: %.0f Mrays/s  %.4f ms/pulse  host %s" % (d["value"], d["ms_per_step"], {k: round(v, 3) for k, v in d["config"]["host_ms_per_pulse_rank0"].items()}))
PY
done
done
cat gpurun_out/${T}.log
