cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out; L=gpurun_out/r04_refr_three_waves_ab.log; : > $L
for rep in 1 2; do for lib in variants/librts_before.so rts_amd/librts_amd.so; do echo "$lib: $(RTS_AMD_LIB=$lib python tools/r04_refr_ab.py 2>&1 | tail -1)" | tee -a $L; done; done
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "refraction or adapter" 2>&1 | tail -2 | tee -a $L
