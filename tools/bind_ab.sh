# pipelined bench with the process on the GPU's socket, on the other socket, and left to the scheduler
cd "${GRAFT_REPO_ROOT:?}"
T=${1:-bind_ab}
NODE=$(rocm-smi --showtoponuma 2>/dev/null | grep "Numa Node:" | head -1 | sed 's/.*: //')
OTHER=$((1 - NODE))
LOCAL=$(cat /sys/devices/system/node/node${NODE}/cpulist); REMOTE=$(cat /sys/devices/system/node/node${OTHER}/cpulist)
echo "GPU on node $NODE (cpus $LOCAL); other node cpus $REMOTE" >> gpurun_out/${T}.log
run() { python3 - "$1" <<PY >> gpurun_out/${T}.log
import json,sys
d=json.loads(open("gpurun_out/${T}_tmp.json").read().strip().splitlines()[-1])
print("%-28s %.0f Mrays/s  %.4f ms/pulse  node %s  host %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["config"].get("host_numa_node"), {k: round(v, 3) for k, v in d["config"]["host_ms_per_pulse_rank0"].items()}))
PY
}
for rep in 1 2 3; do
  python3 bench.py --steps 128 --warmup 8 --no-cpu-baseline > gpurun_out/${T}_tmp.json 2>/dev/null; run "bound (default)"
  python3 bench.py --steps 128 --warmup 8 --no-cpu-baseline --no-bind > gpurun_out/${T}_tmp.json 2>/dev/null; run "scheduler's choice"
  taskset -c $REMOTE python3 bench.py --steps 128 --warmup 8 --no-cpu-baseline --no-bind > gpurun_out/${T}_tmp.json 2>/dev/null; run "other socket (taskset)"
  taskset -c $LOCAL python3 bench.py --steps 128 --warmup 8 --no-cpu-baseline --no-bind > gpurun_out/${T}_tmp.json 2>/dev/null; run "GPU's socket (taskset)"
done
for il in "--inflight 1"; do
  python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline $il > gpurun_out/${T}_tmp.json 2>/dev/null; run "bound, $il"
  python3 bench.py --steps 64 --warmup 8 --no-cpu-baseline $il --no-bind > gpurun_out/${T}_tmp.json 2>/dev/null; run "scheduler, $il"
done
rm -f gpurun_out/${T}_tmp.json
cat gpurun_out/${T}.log
