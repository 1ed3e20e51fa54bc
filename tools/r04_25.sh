cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
T=r04x
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3 4 5; do
  rocprofv3 --hip-trace --output-format csv -d $ROOT/gpurun_out/hiptrace_${T}_$i -- python3 $ROOT/bench.py --steps 128 --no-cpu-baseline > $ROOT/gpurun_out/${T}_bench_$i.json 2> $ROOT/gpurun_out/${T}_bench_$i.err
  python3 $ROOT/tools/bench_line.py $ROOT/gpurun_out/${T}_bench_$i.json | cut -c1-220
  python3 $ROOT/tools/hip_trace_stats.py $ROOT/gpurun_out/hiptrace_${T}_$i > $ROOT/gpurun_out/${T}_hipstats_$i.log; head -9 $ROOT/gpurun_out/${T}_hipstats_$i.log
  rm -rf $ROOT/gpurun_out/hiptrace_${T}_$i
done
