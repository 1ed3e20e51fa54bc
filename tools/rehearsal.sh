# the N > 1 path of bench.py rehearsed on ONE GPU: ranks share the device, collectives over gloo (what runs over RCCL on a multi-GPU node).
# Every combination must complete with every pulse's responses on every rank; the rates mean nothing on a shared GPU.
# usage: tools/rehearsal.sh <tag>   ->  gpurun_out/<tag>_rehearsal_gloo.log
cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
T=${1:-r05}
L=gpurun_out/${T}_rehearsal_gloo.log; : > $L
run() { n=$1; shift; port=$((29540 + RANDOM % 200))
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $port bench.py --gpus $n --backend gloo --no-cpu-baseline "$@" > gpurun_out/${T}x_reh.json 2> gpurun_out/${T}x_reh.err || { echo "FAILED: $n ranks $*" | tee -a $L; tail -5 gpurun_out/${T}x_reh.err | tee -a $L; }
  echo "$n ranks $*: $(python -c "
import json
lines=open('gpurun_out/${T}x_reh.json').read().strip().splitlines(); assert len(lines)==1, 'stdout must hold the JSON line only: %d lines' % len(lines)
j=json.loads(lines[0]); c=j['config']
sec=' | '.join('%s %s' % (k, (round(j[k]['value']) if isinstance(j[k], dict) else round(j[k],3))) for k in ('weak','strong','n1','efficiency_vs_n1') if k in j)
print(round(j['value']), 'Mrays/s', round(j['ms_per_step'],4), 'ms/step n_gpus', j['n_gpus'], 'scaling', j['scaling'], '|', sec, '| sharding:', str(c.get('sharding'))[:110], '| deal:', str(c.get('deal'))[:50])")" | tee -a $L
}
run 1 --steps 20 --warmup 5
run 2 --steps 20 --warmup 5
run 4 --steps 10 --warmup 3
run 2 --steps 21 --warmup 5
run 2 --steps 20 --warmup 5 --scaling weak
run 2 --steps 10 --warmup 6 --shard rays --no-secondary
run 3 --steps 10 --warmup 6 --shard rays --deal interleave --no-secondary
run 2 --config c4 --steps 3 --warmup 3
