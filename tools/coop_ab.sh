# same-box A/B of the cooperative kernel (RTS_COOP_FRAC=0: off) on C3, the dense control and C4 (whole pulse and one GPU's interleaved eighth)
cd "${GRAFT_REPO_ROOT:?}"
T=${1:-coop_ab}
for w in c3 c3narrow c4; do
  for f in 0 0.5 0.25; do
    echo "frac $f $w: $(RTS_COOP_FRAC=$f timeout -k 10 300 python3 tools/trace_bench.py $w 10 | tail -1)" >> gpurun_out/${T}.log
  done
done
for part in 0 7; do
  for f in 0 0.5 0.25; do
    echo "shard 8 part $part frac $f c4: $(RTS_SHARD=8 RTS_SHARD_PART=$part RTS_COOP_FRAC=$f timeout -k 10 300 python3 tools/trace_bench.py c4 8 | tail -1)" >> gpurun_out/${T}.log
  done
done
cat gpurun_out/${T}.log
