# same-box A/B: variants/librts_before.so vs the tree's build.  usage: tools/ab.sh <tag>
cd "${GRAFT_REPO_ROOT:?}"
T=${1:-ab}
for w in c3 c3narrow c3empty c3 c3narrow; do
  echo "before $w: $(RTS_AMD_LIB=variants/librts_before.so python3 tools/trace_bench.py $w 12 | tail -1)" >> gpurun_out/${T}.log
  echo "after  $w: $(python3 tools/trace_bench.py $w 12 | tail -1)" >> gpurun_out/${T}.log
done
RTS_AMD_LIB=variants/librts_before.so python3 bench.py --no-cpu-baseline --steps 64 > gpurun_out/${T}_bench_before.json 2>/dev/null
python3 bench.py --no-cpu-baseline --steps 64 > gpurun_out/${T}_bench_after.json 2>/dev/null
python3 - <<PY >> gpurun_out/${T}.log
import json
for n in ("before","after"):
    d=json.loads(open("gpurun_out/${T}_bench_%s.json"%n).read().strip().splitlines()[-1]); print("bench",n,round(d["value"]),d["ms_per_step"])
PY
cat gpurun_out/${T}.log
