# the bimodal pipelined rate (in a third to a half of the processes the trace launch call blocks ~47 us):
# runtime dispatch settings that could be behind it, each in fresh processes, taken in turn
cd "${GRAFT_REPO_ROOT:?}"
T=${1:-dispatch_ab}
REPS=${2:-6}
run() { python3 - "$1" <<PY >> gpurun_out/${T}.log
import json,sys
try:
    d=json.loads(open("gpurun_out/${T}_tmp.json").read().strip().splitlines()[-1])
    print("%-44s %.0f Mrays/s  %.4f ms/pulse  host %s" % (sys.argv[1], d["value"], d["ms_per_step"], {k: round(v, 3) for k, v in d["config"]["host_ms_per_pulse_rank0"].items()}))
except Exception as e:
    print("%-44s failed: %s" % (sys.argv[1], e))
PY
}
for rep in $(seq 1 $REPS); do
  for v in "X=0" "AMD_DIRECT_DISPATCH=0" "ROC_SIGNAL_POOL_SIZE=512" "AMD_DIRECT_DISPATCH=0 ROC_SIGNAL_POOL_SIZE=512" "ROC_ACTIVE_WAIT_TIMEOUT=1000"; do
    env $v python3 bench.py --steps 128 --warmup 8 --no-cpu-baseline > gpurun_out/${T}_tmp.json 2>/dev/null; run "$v"
  done
done
rm -f gpurun_out/${T}_tmp.json
cat gpurun_out/${T}.log
