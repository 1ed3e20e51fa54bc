cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
L=gpurun_out/r04_coop_big_inflight1.log; : > $L
for cfg in "c3 64" "c5 64" "c2 64" "sphere6 64" "c4 16"; do set -- $cfg
for big in "" 1.5 1.5 ""; do
  RTS_COOP_BIG=$big python bench.py --no-cpu-baseline --config $1 --inflight 1 --steps $2 --warmup 8 > gpurun_out/r04x_b.json 2>/dev/null
  echo "$1 --inflight 1 RTS_COOP_BIG='$big': $(python tools/bench_line.py gpurun_out/r04x_b.json | cut -c1-95)" | tee -a $L
done; done
