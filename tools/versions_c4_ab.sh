# the octant versions on BASELINE configs[3] (1.45 M node records: 1.5 GB of versions) and configs[4]: same-box A/B
cd "${GRAFT_REPO_ROOT:?}"
T=${1:-versions_c4_ab}
: > gpurun_out/${T}.log
for w in c4 c5 c4; do
  echo "sorted   $w: $(RTS_WALK_VERSIONS=0 python3 tools/trace_bench.py $w 10 | tail -1)" >> gpurun_out/${T}.log
  echo "versions $w: $(python3 tools/trace_bench.py $w 10 | tail -1)" >> gpurun_out/${T}.log
done
cat gpurun_out/${T}.log
