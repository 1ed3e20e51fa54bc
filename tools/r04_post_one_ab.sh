cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
L=gpurun_out/r04_post_one_ab.log; : > $L
for rep in 1 2; do for v in 1 0; do
  for cfg in "c3 --inflight 1 --steps 64" "c2 --inflight 1 --steps 64" "sphere6 --inflight 1 --steps 64" "c5 --inflight 1 --steps 64" "c3 --steps 128 --fused-post" "c3 --steps 128"; do
    RTS_POST_ONE=$v python bench.py --no-cpu-baseline --config $cfg > gpurun_out/r04x_po.json 2>/dev/null
    echo "RTS_POST_ONE=$v --config $cfg: $(python tools/bench_line.py gpurun_out/r04x_po.json | cut -c15-90)" | tee -a $L
  done; done; done
