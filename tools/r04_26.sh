cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for stop in 1 2 3 4 0; do
  RTS_AMD_LIB=$ROOT/variants/librts_r04_poststop.so RTS_POST_STOP=$stop rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/gpurun_out/prof_poststop_$stop -- python3 $ROOT/bench.py --steps 64 --inflight 1 --no-cpu-baseline > /dev/null 2>&1
  echo "stop $stop: $(grep k_post_all $(find $ROOT/gpurun_out/prof_poststop_$stop -name '*kernel_stats.csv' | head -1) | cut -d, -f1-7 | cut -c1-200)"
  rm -rf $ROOT/gpurun_out/prof_poststop_$stop
done
