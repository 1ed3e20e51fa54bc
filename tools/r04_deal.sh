cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dealt or interleaved or ragged" 2>&1 | tail -5
for big in 1.5 1.0; do
echo "== RTS_COOP_BIG=$big"
RTS_COOP_BIG=$big timeout -k 10 300 python tools/deal_bench.py c4 8 4096 > gpurun_out/r04_c4_deal_big$big.log 2>&1; tail -2 gpurun_out/r04_c4_deal_big$big.log | cut -c1-300
RTS_COOP_BIG=$big python bench.py --no-cpu-baseline --steps 64 > gpurun_out/r04x_c3_big$big.json 2>/dev/null; python tools/bench_line.py gpurun_out/r04x_c3_big$big.json | cut -c1-120
RTS_COOP_BIG=$big python bench.py --no-cpu-baseline --config c4 --steps 24 --warmup 12 > gpurun_out/r04x_c4_big$big.json 2>/dev/null; python tools/bench_line.py gpurun_out/r04x_c4_big$big.json | cut -c1-120
done
python bench.py --no-cpu-baseline --steps 64 > gpurun_out/r04x_c3_big0.json 2>/dev/null; python tools/bench_line.py gpurun_out/r04x_c3_big0.json | cut -c1-120
