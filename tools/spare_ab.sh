# pipelined bench against the block slots a trace launch leaves free when it shares the GPU (RTS_GRID_SPARE)
cd "${GRAFT_REPO_ROOT:?}"
T=${1:-spare_ab}
for rep in 1 2; do
for sp in ${2:-0 64 128 160 192 256}; do
  RTS_GRID_SPARE=$sp python3 bench.py --steps 128 --warmup 8 --no-cpu-baseline > gpurun_out/${T}_$sp.json 2>/dev/null
  python3 - <<PY >> gpurun_out/${T}.log
import json
d=json.loads(open("gpurun_out/${T}_$sp.json").read().strip().splitlines()[-1])
print("spare $sp: %.0f Mrays/s  %.4f ms/pulse  host %s" % (d["value"], d["ms_per_step"], {k: round(v, 3) for k, v in d["config"]["host_ms_per_pulse_rank0"].items()}))
PY
done
done
cat gpurun_out/${T}.log
