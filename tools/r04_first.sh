# round 4, first GPU call: the suite on the round's starting sources, the C++ boundary with lap timers, the new bench configurations
cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
T=r04a
python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gputests.log 2>&1; tail -3 gpurun_out/${T}_gputests.log
tools/adapter_bench_bin 216 256 3 6 6 3 dh > gpurun_out/${T}_adapter_bench.json 2> gpurun_out/${T}_adapter_bench.err; tail -c 1500 gpurun_out/${T}_adapter_bench.json
python bench.py --no-cpu-baseline --config sphere6 --steps 256 > gpurun_out/${T}_bench_sphere6.json 2> gpurun_out/${T}_bench_sphere6.err; python tools/bench_line.py gpurun_out/${T}_bench_sphere6.json
python bench.py --no-cpu-baseline --config sphere6 --steps 256 --inflight 1 > gpurun_out/${T}_bench_sphere6_inflight1.json 2>/dev/null; python tools/bench_line.py gpurun_out/${T}_bench_sphere6_inflight1.json
python bench.py --no-cpu-baseline --steps 20 --warmup 5 > gpurun_out/${T}_bench_c3_steps20.json 2>/dev/null; python tools/bench_line.py gpurun_out/${T}_bench_c3_steps20.json
python bench.py --no-cpu-baseline --config c5 --steps 1024 --warmup 16 > gpurun_out/${T}_bench_c5.json 2> gpurun_out/${T}_bench_c5.err; python tools/bench_line.py gpurun_out/${T}_bench_c5.json
python bench.py --no-cpu-baseline --config c4 --tx both --steps 24 --warmup 12 > gpurun_out/${T}_bench_c4_2tx.json 2> gpurun_out/${T}_bench_c4_2tx.err; python tools/bench_line.py gpurun_out/${T}_bench_c4_2tx.json
