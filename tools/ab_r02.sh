# same-box A/B: the round-2 final library (variants/librts_r02.so, built from 480ef7b) against the tree's build (and the tree's build without cooperative units)
cd "${GRAFT_REPO_ROOT:?}"
T=${1:-ab_r02}
for rep in 1 2; do
for w in c3 c3narrow; do
  echo "r02 $w: $(RTS_AMD_LIB=variants/librts_r02.so python3 tools/trace_bench.py $w 12 | tail -1)" >> gpurun_out/${T}.log
  echo "now $w: $(python3 tools/trace_bench.py $w 12 | tail -1)" >> gpurun_out/${T}.log
  echo "now, no coop $w: $(RTS_COOP_FRAC=0 python3 tools/trace_bench.py $w 12 | tail -1)" >> gpurun_out/${T}.log
done
for st in "64 8" "20 5"; do
set -- $st
RTS_AMD_LIB=variants/librts_r02.so python3 bench.py --no-cpu-baseline --steps $1 --warmup $2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r02 bench steps $1', round(d['value']), round(d['ms_per_step'],3))" >> gpurun_out/${T}.log
python3 bench.py --no-cpu-baseline --steps $1 --warmup $2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('now bench steps $1', round(d['value']), round(d['ms_per_step'],3), d['config']['interval_tail_ms_rank0'])" >> gpurun_out/${T}.log
RTS_COOP_FRAC=0 python3 bench.py --no-cpu-baseline --steps $1 --warmup $2 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('now, no coop bench steps $1', round(d['value']), round(d['ms_per_step'],3))" >> gpurun_out/${T}.log
done
done
echo "c4: $(python3 tools/trace_bench.py c4 8 | tail -1)" >> gpurun_out/${T}.log
cat gpurun_out/${T}.log
