cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
T=r04t
for tile in 64 512; do for r in 0 1 2 3 4 5 6 7; do echo "tile $tile part $r/8: $(RTS_SHARD_TILE=$tile RTS_SHARD=8 RTS_SHARD_PART=$r RTS_VERBOSE=1 python tools/trace_bench.py c4 7 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-110)" >> gpurun_out/${T}_c4_eighths_tile.log; done; done
cat gpurun_out/${T}_c4_eighths_tile.log
