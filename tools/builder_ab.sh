# same-box comparison of the hierarchies: host SAH, device binned SAH, device LBVH
cd "${GRAFT_REPO_ROOT:?}"
T=${1:-builder_ab}
for w in c3 c3narrow c4; do
  echo "host sah   $w: $(python3 tools/trace_bench.py $w 10 | tail -1)" >> gpurun_out/${T}.log
  echo "device sah $w: $(RTS_BUILDER=device python3 tools/trace_bench.py $w 10 | tail -1)" >> gpurun_out/${T}.log
  echo "device lbvh $w: $(RTS_BUILDER=device RTS_DEVICE_TREE=lbvh python3 tools/trace_bench.py $w 10 | tail -1)" >> gpurun_out/${T}.log
done
cat gpurun_out/${T}.log
