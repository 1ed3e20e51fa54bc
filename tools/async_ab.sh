# same-box A/B of the asynchronous-bounce kernel (rts_trace_unit_async; RTS_ASYNC_IDLE0=0: the lock-step kernel) on C3, the
# dense control, C2 and C4:  idle0 / idle1 = idle-lane limit of a walk phase for young / old tiles, age in cost units (37.5 per us)
cd "${GRAFT_REPO_ROOT:?}"
T=${1:-async_ab}
SETS=${2:-"0:8:7500 64:64:7500 64:8:7500 64:8:3750 64:16:7500 64:4:7500 32:8:7500 16:8:7500 8:8:0"}
WL=${3:-"c3 c3narrow c2"}
for w in $WL; do
  for s in $SETS; do
    i0=${s%%:*}; r=${s#*:}; i1=${r%%:*}; ag=${r#*:}
    echo "idle0 $i0 idle1 $i1 age $ag $w: $(RTS_ASYNC_IDLE0=$i0 RTS_ASYNC_IDLE1=$i1 RTS_ASYNC_AGE=$ag timeout -k 10 300 python3 tools/trace_bench.py $w 12 | tail -1)" >> gpurun_out/${T}.log
  done
done
cat gpurun_out/${T}.log
