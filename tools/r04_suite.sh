cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout -k 10 300 python tools/deal_bench.py c4 8 4096 > gpurun_out/r04_c4_deal.log 2>&1; tail -3 gpurun_out/r04_c4_deal.log | cut -c1-300
