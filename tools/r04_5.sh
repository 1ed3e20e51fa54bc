cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
python -m cProfile -s tottime bench.py --config sphere6 --no-cpu-baseline --steps 256 2>/dev/null | grep -v "^{" | head -30 > gpurun_out/r04e_cprofile_sphere6.txt; cat gpurun_out/r04e_cprofile_sphere6.txt
