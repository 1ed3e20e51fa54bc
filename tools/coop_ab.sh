# same-box A/B of the cooperative kernel (RTS_COOP_FRAC=0: off) on C3, the dense control and C4
cd "${GRAFT_REPO_ROOT:?}"
T=${1:-coop_ab}
for w in c3 c3narrow; do
  for f in 0 0.5; do
    echo "frac $f $w: $(RTS_COOP_FRAC=$f timeout -k 10 300 python3 tools/trace_bench.py $w 10 | tail -1)" >> gpurun_out/${T}.log
  done
done
for f in 0 0.5 0.75 1.0 0.35; do
  echo "frac $f c4: $(RTS_COOP_FRAC=$f timeout -k 10 300 python3 tools/trace_bench.py c4 8 | tail -1)" >> gpurun_out/${T}.log
done
for g in 256 512 2048; do
  echo "frac 0.5 grid $g c4: $(RTS_COOP_GRID=$g timeout -k 10 300 python3 tools/trace_bench.py c4 8 | tail -1)" >> gpurun_out/${T}.log
done
echo "shard 8 frac 0 c4: $(RTS_SHARD=8 RTS_COOP_FRAC=0 timeout -k 10 300 python3 tools/trace_bench.py c4 8 | tail -1)" >> gpurun_out/${T}.log
echo "shard 8 frac 0.5 c4: $(RTS_SHARD=8 timeout -k 10 300 python3 tools/trace_bench.py c4 8 | tail -1)" >> gpurun_out/${T}.log
cat gpurun_out/${T}.log
