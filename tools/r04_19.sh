cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
RTS_DEBUG_COOP=1 RTS_SHARD=8 RTS_SHARD_PART=6 RTS_TIMELINE_LAUNCHES=5 python tools/timeline.py c4 > gpurun_out/r04r_timeline_c4_part6.log 2>&1; cut -c1-400 gpurun_out/r04r_timeline_c4_part6.log
