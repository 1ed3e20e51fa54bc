# the N > 1 path of bench.py rehearsed on ONE GPU: ranks share the device, collectives over gloo (what runs over RCCL on a multi-GPU node)
cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
L=gpurun_out/r04_rehearsal_gloo.log; : > $L
run() { n=$1; shift; port=$((29540 + RANDOM % 200))
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port bench.py --gpus $n --backend gloo --no-cpu-baseline "$@" > gpurun_out/r04x_reh.json 2> gpurun_out/r04x_reh.err || { echo "FAILED: $n ranks $*"; tail -5 gpurun_out/r04x_reh.err; }
  echo "$n ranks $*: $(python -c "
import json
j=json.loads(open('gpurun_out/r04x_reh.json').read().strip().splitlines()[-1]); c=j['config']
print(round(j['value']), 'Mrays/s', round(j['ms_per_step'],4), 'ms/pulse n_gpus', j['n_gpus'], 'scaling', j['scaling'], '| responses', c.get('responses_per_pulse', c.get('responses')), '| sharding:', str(c.get('sharding'))[:90], '| deal:', str(c.get('deal'))[:60])")" | tee -a $L
}
run 1 --steps 20 --warmup 5
run 2 --steps 20 --warmup 5
run 4 --steps 10 --warmup 3
run 2 --steps 20 --warmup 5 --scaling strong
run 2 --steps 21 --warmup 5 --scaling strong
run 2 --steps 10 --warmup 6 --shard rays
run 3 --steps 10 --warmup 6 --shard rays --deal cost
run 2 --config c4 --steps 3 --warmup 3 --shard rays --deal cost
