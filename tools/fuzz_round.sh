# differential fuzz of a round's build: standard + big scenes against the oracle, then the same generator with every tile position screened
# tile-wise first (RTS_DEAD_BATCH=all) and with a small grid (cost orders on small launches).  usage: tools/fuzz_round.sh <tag> <n> <seed0> <n_big>
cd "${GRAFT_REPO_ROOT:?}"; T=${1:-fuzz}; N=${2:-600}; S=${3:-600000}; NB=${4:-100}
timeout -k 10 900 python tools/fuzz_equal.py $N $S --oracle > gpurun_out/${T}_standard.log 2>&1; tail -2 gpurun_out/${T}_standard.log
timeout -k 10 600 python tools/fuzz_equal.py $NB $((S+100000)) --big --oracle > gpurun_out/${T}_big.log 2>&1; tail -2 gpurun_out/${T}_big.log
RTS_DEAD_BATCH=all timeout -k 10 600 python tools/fuzz_equal.py $((N/2)) $((S+200000)) --oracle > gpurun_out/${T}_batch_all.log 2>&1; tail -2 gpurun_out/${T}_batch_all.log
RTS_DEAD_BATCH=all RTS_GRID_MULT=1 timeout -k 10 600 python tools/fuzz_equal.py $((NB/2)) $((S+300000)) --big --oracle > gpurun_out/${T}_batch_all_big.log 2>&1; tail -2 gpurun_out/${T}_batch_all_big.log
