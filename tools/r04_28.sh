cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
for st in 600 1000 1500 2500; do
  RTS_COOP_STEPS=$st python bench.py --no-cpu-baseline --config c5 --steps 512 --warmup 16 > gpurun_out/r04z_c5.json 2>/dev/null; echo "steps $st c5: $(python tools/bench_line.py gpurun_out/r04z_c5.json | cut -c1-110)"
  for r in 3 6; do echo "steps $st c4 part $r/8: $(RTS_COOP_STEPS=$st RTS_SHARD=8 RTS_SHARD_PART=$r RTS_VERBOSE=1 python tools/trace_bench.py c4 7 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-100)"; done
  echo "steps $st c4 whole: $(RTS_COOP_STEPS=$st RTS_VERBOSE=1 python tools/trace_bench.py c4 8 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-110)"
done
