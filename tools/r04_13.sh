cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
T=r04l
for v in old new old new old new; do
  if [ $v = old ]; then L=$PWD/variants/librts_r04_before_xcd.so; else L=$PWD/rts_amd/librts_amd.so; fi
  echo "$v: $(RTS_AMD_LIB=$L RTS_VERBOSE=1 python tools/trace_bench.py c4 12 2>&1 | tail -2 | tr '\n' ' ')" >> gpurun_out/${T}_c4_coop_xcd_ab.log
done
cut -c1-360 gpurun_out/${T}_c4_coop_xcd_ab.log
for v in old new; do
  if [ $v = old ]; then L=$PWD/variants/librts_r04_before_xcd.so; else L=$PWD/rts_amd/librts_amd.so; fi
  RTS_AMD_LIB=$L python bench.py --no-cpu-baseline --config c4 --steps 24 --warmup 12 > gpurun_out/${T}_bench_c4_$v.json 2>/dev/null; python tools/bench_line.py gpurun_out/${T}_bench_c4_$v.json
done
