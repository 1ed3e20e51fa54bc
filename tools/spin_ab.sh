# the pulse's host waits: polling the stream (RTS_SPIN_WAIT=1, default) against blocking in hipStreamSynchronize
cd "${GRAFT_REPO_ROOT:?}"
T=${1:-spin_ab}
for rep in 1 2 3; do
for v in 1 0; do
  RTS_SPIN_WAIT=$v python3 bench.py --steps 128 --warmup 8 --no-cpu-baseline > gpurun_out/${T}_a.json 2>/dev/null
  RTS_SPIN_WAIT=$v python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline --inflight 1 > gpurun_out/${T}_b.json 2>/dev/null
  python3 - <<PY >> gpurun_out/${T}.log
import json
a=json.loads(open("gpurun_out/${T}_a.json").read().strip().splitlines()[-1]); b=json.loads(open("gpurun_out/${T}_b.json").read().strip().splitlines()[-1])
print("spin $v: pipelined %.0f Mrays/s %.4f ms/pulse host %s | one at a time %.4f ms/pulse" % (a["value"], a["ms_per_step"], {k: round(x, 3) for k, x in a["config"]["host_ms_per_pulse_rank0"].items()}, b["ms_per_step"]))
PY
  echo "spin $v adapter: $(RTS_SPIN_WAIT=$v tools/adapter_bench_bin 216 64 3 6 6 2>&1 | tail -1 | cut -c1-330)" >> gpurun_out/${T}.log
  echo "spin $v adapter, host tree: $(RTS_SPIN_WAIT=$v RTS_BUILDER=host tools/adapter_bench_bin 216 64 3 6 6 2>&1 | tail -1 | cut -c1-330)" >> gpurun_out/${T}.log
done
done
rm -f gpurun_out/${T}_a.json gpurun_out/${T}_b.json
cat gpurun_out/${T}.log
