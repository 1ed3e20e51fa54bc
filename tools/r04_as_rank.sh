cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
L=gpurun_out/r04_c4_as_rank_deal.log; : > $L
for deal in cost interleave; do
for r in 0 1 2 3 4 5 6 7; do
  python bench.py --no-cpu-baseline --config c4 --shard rays --as-rank $r/8 --deal $deal --steps 48 --warmup 12 > gpurun_out/r04x_asrank.json 2> gpurun_out/r04x_asrank.err || tail -3 gpurun_out/r04x_asrank.err
  echo "--deal $deal rank $r/8: $(python tools/bench_line.py gpurun_out/r04x_asrank.json | cut -c1-110) $(python -c "
import json; j=json.loads(open('gpurun_out/r04x_asrank.json').read().strip().splitlines()[-1]); d=j['config']['deal']; print('cost share', d['cost'] if isinstance(d, dict) else d)")" | tee -a $L
done
done
