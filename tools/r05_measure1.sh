# round 5, after the octant versions / dead-tile batches / window screen: lane statistics with the never-started split, the lone configs[3]
# launch's timeline, configs[3] parts per rank, the headline both ways.  usage: tools/r05_measure1.sh
cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
python3 tools/count_stats.py c3 c3narrow c2 c4 c5 > gpurun_out/r05c_count_stats.log 2>&1; cat gpurun_out/r05c_count_stats.log
RTS_TIMELINE_LAUNCHES=4 python3 tools/timeline.py c4 > gpurun_out/r05c_timeline_c4.log 2>&1; tail -14 gpurun_out/r05c_timeline_c4.log
python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/r05c_bench_c3_steps20.json 2>/dev/null; python3 tools/bench_line.py gpurun_out/r05c_bench_c3_steps20.json | cut -c1-150
python3 bench.py --no-cpu-baseline > gpurun_out/r05c_bench_c3_256.json 2>/dev/null; python3 tools/bench_line.py gpurun_out/r05c_bench_c3_256.json | cut -c1-150
python3 bench.py --no-cpu-baseline --config c2 > gpurun_out/r05c_bench_c2.json 2>/dev/null; python3 tools/bench_line.py gpurun_out/r05c_bench_c2.json | cut -c1-150
python3 bench.py --no-cpu-baseline --config c5 --steps 1024 > gpurun_out/r05c_bench_c5.json 2>/dev/null; python3 tools/bench_line.py gpurun_out/r05c_bench_c5.json | cut -c1-150
bash tools/as_rank.sh r05c_c4_as_rank > /dev/null 2>&1; tail -30 gpurun_out/r05c_c4_as_rank.log | cut -c1-250
