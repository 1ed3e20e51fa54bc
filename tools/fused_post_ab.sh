# rts_trace_pulse_end_uniform (speculative post-processing on the device-side received count; on the handle's own stream or behind
# the trace kernel on its stream) against the four calls with the host waiting for the trace in between
cd "${GRAFT_REPO_ROOT:?}"
T=${1:-fused_post_ab}
for rep in 1 2 3; do
for v in four own trace; do
  a="--fused-post"; if [ $v = four ]; then a=""; fi
  RTS_SPEC_STREAM=$v python3 bench.py --steps 128 --warmup 8 --no-cpu-baseline $a > gpurun_out/${T}_a.json 2>/dev/null
  RTS_SPEC_STREAM=$v python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline --inflight 1 $a > gpurun_out/${T}_b.json 2>/dev/null
  python3 - <<PY >> gpurun_out/${T}.log
import json
a=json.loads(open("gpurun_out/${T}_a.json").read().strip().splitlines()[-1]); b=json.loads(open("gpurun_out/${T}_b.json").read().strip().splitlines()[-1])
print("%-6s pipelined %.0f Mrays/s %.4f ms/pulse host %s | one at a time %.4f ms/pulse" % ("$v", a["value"], a["ms_per_step"], {k: round(x, 3) for k, x in a["config"]["host_ms_per_pulse_rank0"].items()}, b["ms_per_step"]))
PY
done
done
rm -f gpurun_out/${T}_a.json gpurun_out/${T}_b.json
cat gpurun_out/${T}.log
