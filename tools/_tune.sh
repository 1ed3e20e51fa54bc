cd $GRAFT_REPO_ROOT
run() { python3 bench.py --no-cpu-baseline --steps 64 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']), round(j['ms_per_step'],4), round(j['roofline']['kernel_ms_serial'],3))"; }
for s in 64 128 192 256 384 512 128 256; do echo "spare$s $(RTS_GRID_SPARE=$s run)" >> gpurun_out/tune.log; done
echo "spare256 inflight2 $(RTS_GRID_SPARE=256 run --inflight 2)" >> gpurun_out/tune.log
echo "spare256 inflight4 $(RTS_GRID_SPARE=256 run --inflight 4)" >> gpurun_out/tune.log
cat gpurun_out/tune.log
