cd "${GRAFT_REPO_ROOT:?}"
T=${1:-bind_ab2}
NODE=$(rocm-smi --showtoponuma 2>/dev/null | grep "Numa Node:" | head -1 | sed 's/.*: //')
B=$((NODE * 64)); S=$((B + 128))
echo "GPU on node $NODE" >> gpurun_out/${T}.log
run() { python3 - "$1" <<PY >> gpurun_out/${T}.log
import json,sys
d=json.loads(open("gpurun_out/${T}_tmp.json").read().strip().splitlines()[-1])
print("%-34s %.0f Mrays/s  %.4f ms/pulse  host %s" % (sys.argv[1], d["value"], d["ms_per_step"], {k: round(v, 3) for k, v in d["config"]["host_ms_per_pulse_rank0"].items()}))
PY
}
for rep in 1 2 3 4; do
  for set in "$B-$((B+63)),$S-$((S+63))" "$B-$((B+7)),$S-$((S+7))" "$((B+8))-$((B+15))" "$((B+32))-$((B+39)),$((S+32))-$((S+39))" "$B-$((B+1))"; do
    taskset -c $set python3 bench.py --steps 128 --warmup 8 --no-cpu-baseline --no-bind > gpurun_out/${T}_tmp.json 2>/dev/null; run "taskset $set"
  done
done
rm -f gpurun_out/${T}_tmp.json
cat gpurun_out/${T}.log
