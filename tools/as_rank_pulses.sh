# strong scaling of a SHORT interval (the driver's --steps 20), one GPU running each rank's plan in turn (bench.py --as-rank r/N --shard whole | pulses | rays):
# rank r's wall time for its share against the N = 1 time of the same interval.  usage: tools/as_rank_pulses.sh <tag> <config> [N=8] [steps=20]
cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
T=${1:-as_rank_pulses}; C=${2:-c3}; N=${3:-8}; K=${4:-20}
L=gpurun_out/${T}.log; : > $L
python bench.py --no-cpu-baseline --config $C --steps $K --warmup 5 > gpurun_out/${T}_x.json 2>/dev/null
T1=$(python -c "import json; j=json.loads(open('gpurun_out/${T}_x.json').read()); print(j['ms_per_step']*j['steps'])")
echo "$C, $K pulses on ONE GPU: $T1 ms" | tee -a $L
for shard in whole pulses rays; do
for r in $(seq 0 $((N-1))); do
  python bench.py --no-cpu-baseline --config $C --shard $shard --as-rank $r/$N --steps $K --warmup 5 > gpurun_out/${T}_x.json 2>/dev/null
  echo "--shard $shard rank $r/$N: $(python -c "
import json; j=json.loads(open('gpurun_out/${T}_x.json').read()); t=j['ms_per_step']*j['steps']; print('%.3f ms for its share of the %d-pulse interval -> efficiency if every rank took that long: %.2f' % (t, j['steps'], $T1/($N*t)))")" | tee -a $L
done
done
