# tile cost records on the constant-rate counter.  (1) product build, 512 BASELINE configs[3] pulses, three in flight, RTS_DEBUG_COOP=1:
# the library prints a "clocks:" line for every launch that dropped a cost record -- there must be none; (2) the counting build in the same
# loop reads BOTH clocks per tile and counts the tiles whose shader clock (s_memtime, what rounds 2-4 timed tiles with) ran backwards.
cd "${GRAFT_REPO_ROOT:?}"; T=${1:-r05_clock_check}; L=gpurun_out/${T}.log; : > $L
RTS_DEBUG_COOP=1 timeout -k 10 400 python bench.py --config c4 --steps 512 --warmup 12 --no-cpu-baseline > gpurun_out/${T}_product.json 2> gpurun_out/${T}_product.err
echo "product build, 512 pulses: launches $(grep -c '\[rts\] launch:' gpurun_out/${T}_product.err), 'clocks:' lines (records dropped / shader clock backwards): $(grep -c 'clocks:' gpurun_out/${T}_product.err)" >> $L
python -c "
import json; j=json.loads(open('gpurun_out/${T}_product.json').read().strip().splitlines()[-1]); print('   ', round(j['value']), 'Mrays/s', round(j['ms_per_step'],3), 'ms/pulse')" >> $L
RTS_DEBUG_COOP=1 RTS_BENCH_COUNT=1 timeout -k 10 400 python bench.py --config c4 --steps 256 --warmup 12 --no-cpu-baseline > gpurun_out/${T}_count.json 2> gpurun_out/${T}_count.err
echo "counting build, 256 pulses: launches $(grep -c '\[rts\] launch:' gpurun_out/${T}_count.err), 'clocks:' lines: $(grep -c 'clocks:' gpurun_out/${T}_count.err)" >> $L
grep 'clocks:' gpurun_out/${T}_count.err | sort | uniq -c | sort -rn | head -12 >> $L
cat $L
