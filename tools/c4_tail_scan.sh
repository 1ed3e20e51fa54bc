# BASELINE configs[3] after the window screen: a lone launch is 7.2 ms against a balanced time of ~2 ms -- a dozen tiles of 4-8 ms that the cooperative
# kernel's head rule does not take.  Scan of the rule's knobs (lone launches: tools/trace_bench.py c4; pipelined: bench.py --inflight).  usage: tools/c4_tail_scan.sh <tag>
cd "${GRAFT_REPO_ROOT:?}"; T=${1:-c4_tail_scan}; L=gpurun_out/${T}.log; : > $L
run() { echo "$*: $(env "$@" RTS_VERBOSE=1 python3 tools/trace_bench.py c4 10 | tail -2 | tr '\n' ' ' | cut -c1-330)" | tee -a $L; }
run RTS_COOP_BIG=0
run RTS_COOP_BIG=2
run RTS_COOP_BIG=1
run RTS_COOP_BIG=0.5
run RTS_COOP_MID=1
run RTS_COOP_MID=0.5 RTS_COOP_STEPS_LO=200
run RTS_COOP_STEPS=400 RTS_COOP_STEPS_LO=100 RTS_COOP_MID=1
run RTS_COOP_FRAC=0.25
for f in 3 4 6 8; do
  python3 bench.py --no-cpu-baseline --config c4 --steps 48 --warmup 16 --inflight $f > gpurun_out/${T}_x.json 2>/dev/null
  echo "--inflight $f: $(python3 tools/bench_line.py gpurun_out/${T}_x.json | cut -c1-100)" | tee -a $L
done
