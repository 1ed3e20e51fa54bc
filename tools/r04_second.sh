# round 4, second GPU call: new API parity, the adapter's staged path
cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
T=r04b
python -m pytest tests -m gpu -x -q -k "host_mirror or wide_keys or pulse_end_uniform or adapter" > gpurun_out/${T}_newtests.log 2>&1; tail -15 gpurun_out/${T}_newtests.log
tools/adapter_bench_bin 216 256 3 6 6 3 dh > gpurun_out/${T}_adapter_bench.json 2> gpurun_out/${T}_adapter_bench.err; python tools/adapter_line.py gpurun_out/${T}_adapter_bench.json; tail -3 gpurun_out/${T}_adapter_bench.err
tools/adapter_bench_bin 216 256 2 6 6 2 d > gpurun_out/${T}_adapter_bench_f2.json 2>/dev/null; python tools/adapter_line.py gpurun_out/${T}_adapter_bench_f2.json
tools/adapter_bench_bin 216 256 1 6 6 2 d > gpurun_out/${T}_adapter_bench_f1.json 2>/dev/null; python tools/adapter_line.py gpurun_out/${T}_adapter_bench_f1.json
python -m pytest tests -m gpu -x -q > gpurun_out/${T}_gputests.log 2>&1; tail -3 gpurun_out/${T}_gputests.log
