# one GPU running, rank by rank, what each of 8 ranks of a ray-sharded BASELINE configs[3] interval would run (bench.py --as-rank r/8):
# pipelined parts (three in flight), dealt by cost and interleaved; then one part at a time (tools/deal_bench.py).  usage: tools/as_rank.sh <tag> [ranks]
cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
T=${1:-as_rank}; RANKS=${2:-"0 1 2 3 4 5 6 7"}
L=gpurun_out/${T}.log; : > $L
for deal in cost interleave; do
for r in $RANKS; do
  python bench.py --no-cpu-baseline --config c4 --shard rays --as-rank $r/8 --deal $deal --steps 48 --warmup 12 > gpurun_out/${T}_x.json 2> gpurun_out/${T}_x.err || tail -3 gpurun_out/${T}_x.err
  echo "--deal $deal rank $r/8: $(python tools/bench_line.py gpurun_out/${T}_x.json | cut -c1-110) $(python -c "
import json; j=json.loads(open('gpurun_out/${T}_x.json').read().strip().splitlines()[-1]); d=j['config']['deal']; print('cost share', d['cost'] if isinstance(d, dict) else d)")" | tee -a $L
done
done
python bench.py --no-cpu-baseline --config c4 --steps 24 --warmup 12 > gpurun_out/${T}_whole.json 2>/dev/null
echo "whole pulses, pipelined: $(python tools/bench_line.py gpurun_out/${T}_whole.json | cut -c1-110)" | tee -a $L
python tools/deal_bench.py c4 8 4096 7 2>&1 | tee -a $L
