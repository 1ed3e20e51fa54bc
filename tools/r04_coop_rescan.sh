cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out
L=gpurun_out/r04_coop_rescan.log; : > $L
for f in 0.25 0.35 0.5 0.75; do echo "RTS_COOP_FRAC=$f c4: $(RTS_COOP_FRAC=$f python3 tools/trace_bench.py c4 10 2>&1 | tail -1 | cut -c60-140)" | tee -a $L; done
for s in 600 1000 1500; do echo "RTS_COOP_STEPS=$s c4: $(RTS_COOP_STEPS=$s python3 tools/trace_bench.py c4 10 2>&1 | tail -1 | cut -c60-140)" | tee -a $L; done
for m in 2 3 5; do echo "RTS_COOP_MID=$m c4: $(RTS_COOP_MID=$m python3 tools/trace_bench.py c4 10 2>&1 | tail -1 | cut -c60-140) | c5: $(RTS_COOP_MID=$m python3 tools/trace_bench.py c5 12 2>&1 | tail -1 | cut -c60-140)" | tee -a $L; done
for b in 1.0 1.5 2.5; do RTS_COOP_BIG_PART=$b python tools/deal_bench.py c4 8 4096 2>&1 | tail -2 | cut -c1-170 | sed "s|^|RTS_COOP_BIG_PART=$b |" | tee -a $L; done
