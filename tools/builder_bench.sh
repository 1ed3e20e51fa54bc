cd "${GRAFT_REPO_ROOT:?}"
T=${1:-builder_bench}
for rep in 1 2 3; do
  for b in host device; do
    RTS_BUILDER=$b RTS_SPLIT_BUDGET=2 python3 bench.py --no-cpu-baseline --steps 64 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$b', round(d['value']), round(d['ms_per_step'],3), 'serial', round(r['kernel_ms_serial'],3), 'dense', round(r['dense_control']['kernel_ms'],3), 'setup_s', round(d['config']['scene_setup_s'],3), 'V,T', round(r['nodes_per_segment'],2), round(r['tri_tests_per_segment'],2))" >> gpurun_out/${T}.log
  done
done
cat gpurun_out/${T}.log
