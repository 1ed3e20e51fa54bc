cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
for v in new nocoop old new nocoop old; do
  case $v in new) E="";; nocoop) E="RTS_COOP_FRAC=0";; old) E="RTS_AMD_LIB=$PWD/variants/librts_r04_before_xcd.so";; esac
  env $E python bench.py --no-cpu-baseline --config c5 --steps 512 --warmup 16 > gpurun_out/r04y_c5_$v.json 2>/dev/null; echo "$v: $(python tools/bench_line.py gpurun_out/r04y_c5_$v.json | cut -c1-230)"
done
RTS_DEBUG_COOP=1 python tools/trace_bench.py c5 6 2>&1 | tail -8 | cut -c1-200
