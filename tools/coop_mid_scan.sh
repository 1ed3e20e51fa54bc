# the head rule's LONGISH threshold (RTS_COOP_MID x the launch's balanced time; 3 since round 4) re-scanned after the dead work left the balanced time,
# with the cooperative records' decay (rts_record_to_keep): lone launches and the pipelined bench on every configuration.  usage: tools/coop_mid_scan.sh <tag>
cd "${GRAFT_REPO_ROOT:?}"; T=${1:-coop_mid_scan}; L=gpurun_out/${T}.log; : > $L
for mid in 3 1.5 1 0.7; do
  for w in c4 c3 c5 c2; do
    echo "RTS_COOP_MID=$mid lone $w: $(RTS_COOP_MID=$mid RTS_VERBOSE=1 python3 tools/trace_bench.py $w 12 | tail -2 | tr '\n' ' ' | cut -c1-300)" | tee -a $L
  done
  for c in c4:32:12 c3:64:8 c5:128:8; do IFS=: read cfg st wu <<< "$c"
    RTS_COOP_MID=$mid python3 bench.py --no-cpu-baseline --config $cfg --steps $st --warmup $wu > gpurun_out/${T}_x.json 2>/dev/null
    echo "RTS_COOP_MID=$mid pipelined $cfg: $(python3 tools/bench_line.py gpurun_out/${T}_x.json | cut -c1-100)" | tee -a $L
  done
done
