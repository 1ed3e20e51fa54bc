cd "${GRAFT_REPO_ROOT:?}"
run() { python3 bench.py --steps 64 --inflight 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value']), round(d['ms_per_step'],3), {k:round(x,3) for k,x in d['config']['stage_ms_per_launch_rank0'].items()}, 'serial', round(d['roofline']['kernel_ms_serial'],3))"; }
for rep in 1 2; do
  RTS_AMD_LIB=variants/librts_r02.so run "r02 inflight1"
  unset RTS_AMD_LIB
  run "now inflight1"
  RTS_BUILDER=host run "now host-tree inflight1"
  RTS_BUILDER=host RTS_COOP_FRAC=0 run "now host-tree no-coop inflight1"
done
for rep in 1 2; do
  python3 bench.py --steps 64 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('now default', round(d['value']), round(d['ms_per_step'],3))"
  RTS_COOP_FRAC=0 python3 bench.py --steps 64 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('now frac0', round(d['value']), round(d['ms_per_step'],3))"
done
