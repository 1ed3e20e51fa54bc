#!/bin/bash
# Same-box A/B between ENVIRONMENT settings (or libraries: RTS_AMD_LIB=variants/x.so is a setting like any other) of one build:
# lone launches (tools/trace_bench.py, HIP events inside the library, checksum of the received set) for every workload x setting, the
# settings interleaved and the whole round repeated, then -- with BENCH="cfg:steps:warmup ..." -- the pipelined bench line per setting.
#   tools/ab_env.sh <tag> "<workloads>" "<setting A>" "<setting B>" [...]        a setting: "NAME=value NAME2=value2", or "-" for none
#   e.g.  BENCH="c3:64:8 c4:24:12" tools/ab_env.sh r05x_batch_ab "c3 c3empty c4" "RTS_DEAD_BATCH=0" "-"
# Writes gpurun_out/<tag>.log, its first line the command.  (Rounds 2-5 each grew a dozen one-off *_ab.sh of this shape; they are gone, the logs
# they produced under profiles/ carry their command in the first line -- profiles/README.md.)
cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
T=$1; W=$2; shift 2
L=gpurun_out/${T}.log
echo "# BENCH=\"${BENCH:-}\" REPS=${REPS:-12} ROUNDS=${ROUNDS:-2} tools/ab_env.sh $T \"$W\" $(printf '"%s" ' "$@")" > $L
for round in $(seq 1 ${ROUNDS:-2}); do for w in $W; do for s in "$@"; do
  [ "$s" = "-" ] && e="" || e="$s"
  echo "[$s] $w: $(env $e RTS_VERBOSE=${VERBOSE:-} python3 tools/trace_bench.py $w ${REPS:-12} | tail -${VERBOSE:+2}${VERBOSE:-1} | tr '\n' ' ' | cut -c1-360)" >> $L
done; done; done
for c in ${BENCH:-}; do IFS=: read cfg st wu <<< "$c"; for s in "$@"; do
  [ "$s" = "-" ] && e="" || e="$s"
  env $e python3 bench.py --no-cpu-baseline --config $cfg --steps $st --warmup $wu ${BENCH_ARGS:-} > gpurun_out/${T}_x.json 2>/dev/null
  echo "[$s] bench --config $cfg --steps $st --warmup $wu ${BENCH_ARGS:-}: $(python3 tools/bench_line.py gpurun_out/${T}_x.json | cut -c1-120)" >> $L
done; done
rm -f gpurun_out/${T}_x.json
cat $L
