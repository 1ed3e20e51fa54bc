cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out; rm -f gpurun_out/r04_prefilter_ab.log
timeout -k 10 600 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
bash tools/ab.sh r04_prefilter_ab > /dev/null 2>&1; cat gpurun_out/r04_prefilter_ab.log | cut -c1-220
RTS_AMD_LIB=variants/librts_before.so python3 bench.py --no-cpu-baseline > gpurun_out/r04x_pf_before.json 2>/dev/null; python3 bench.py --no-cpu-baseline > gpurun_out/r04x_pf_after.json 2>/dev/null
RTS_AMD_LIB=variants/librts_before.so python3 bench.py --no-cpu-baseline > gpurun_out/r04x_pf_before2.json 2>/dev/null; python3 bench.py --no-cpu-baseline > gpurun_out/r04x_pf_after2.json 2>/dev/null
python tools/bench_line.py gpurun_out/r04x_pf_before.json gpurun_out/r04x_pf_after.json gpurun_out/r04x_pf_before2.json gpurun_out/r04x_pf_after2.json | cut -c1-120 | tee -a gpurun_out/r04_prefilter_ab.log
timeout -k 10 400 python tools/fuzz_equal.py 300 8300000 --oracle > gpurun_out/r04_fuzz3_standard.log 2>&1; tail -2 gpurun_out/r04_fuzz3_standard.log | cut -c1-200
