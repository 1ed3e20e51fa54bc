cd $GRAFT_REPO_ROOT
run() { python3 bench.py --no-cpu-baseline --steps 64 "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); print(round(j['value']), round(j['ms_per_step'],4), round(j['roofline']['kernel_ms_serial'],3), round(j['roofline']['dense_control']['kernel_ms'],3))"; }
for v in before after before after; do if [ $v = before ]; then echo "$v $(RTS_AMD_LIB=variants/librts_before.so run)" >> gpurun_out/tune.log; else echo "$v $(run)" >> gpurun_out/tune.log; fi; done
cat gpurun_out/tune.log
