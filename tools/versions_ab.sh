# (profiles/r05a_versions_ab.log was taken when the entry-corner key was the default: its 'versions' rows are the corner key, 'v-centre' the centre)
# same-box A/B of the octant versions of the node records (RTS_WALK_VERSIONS=0: role fetch + sorting network).  usage: tools/versions_ab.sh <tag> [workloads]
cd "${GRAFT_REPO_ROOT:?}"
T=${1:-versions_ab}; shift
W=${@:-c3 c3narrow c2 c3 c3narrow}
: > gpurun_out/${T}.log
for w in $W; do
  echo "sorted   $w: $(RTS_WALK_VERSIONS=0 python3 tools/trace_bench.py $w 12 | tail -1)" >> gpurun_out/${T}.log
  echo "versions $w: $(python3 tools/trace_bench.py $w 12 | tail -1)" >> gpurun_out/${T}.log
  if [ "$RTS_AB_KEYS" ]; then echo "v-corner $w: $(RTS_VERSION_KEY=corner python3 tools/trace_bench.py $w 12 | tail -1)" >> gpurun_out/${T}.log; fi
done
echo "--- counting build (node visits / triangle tests per segment)" >> gpurun_out/${T}.log
echo "sorted:" >> gpurun_out/${T}.log;   RTS_WALK_VERSIONS=0 python3 tools/count_stats.py c3 c3narrow c2 >> gpurun_out/${T}.log
echo "versions (centre, the default):" >> gpurun_out/${T}.log; python3 tools/count_stats.py c3 c3narrow c2 >> gpurun_out/${T}.log
echo "versions (entry corner):" >> gpurun_out/${T}.log; RTS_VERSION_KEY=corner python3 tools/count_stats.py c3 c3narrow c2 >> gpurun_out/${T}.log
RTS_WALK_VERSIONS=0 python3 bench.py --no-cpu-baseline --steps 64 > gpurun_out/${T}_bench_sorted.json 2>/dev/null
python3 bench.py --no-cpu-baseline --steps 64 > gpurun_out/${T}_bench_versions.json 2>/dev/null
python3 - <<PY >> gpurun_out/${T}.log
import json
for n in ("sorted","versions"):
    d=json.loads(open("gpurun_out/${T}_bench_%s.json"%n).read().strip().splitlines()[-1]); print("bench --steps 64",n,round(d["value"]),"Mrays/s",round(d["ms_per_step"],4),"ms/pulse; serial kernel",round(d["roofline"]["kernel_ms_serial"],4),"dense",round(d["roofline"]["dense_control"]["Gseg_per_s"],3))
PY
cat gpurun_out/${T}.log
