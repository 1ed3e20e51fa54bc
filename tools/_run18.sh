cd $GRAFT_REPO_ROOT
for shard in pulses rays; do
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 7 --warmup 2 --backend gloo --no-cpu-baseline --shard $shard > gpurun_out/r02_gloo2_$shard.json 2> gpurun_out/r02_gloo2_$shard.err; echo "rc=$?"; python -c "
import json
j=json.loads(open('gpurun_out/r02_gloo2_$shard.json').read().strip().splitlines()[-1]); print('$shard', j['n_gpus'], j['value'], j['ms_per_step'], j['config']['sharding'][:80], j['config']['interval_tail_ms_rank0'])"
done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 3 --steps 4 --warmup 1 --backend gloo --no-cpu-baseline --shard rays > gpurun_out/r02_gloo3_rays.json 2> gpurun_out/r02_gloo3_rays.err; echo "rc=$?"; tail -c 300 gpurun_out/r02_gloo3_rays.err
