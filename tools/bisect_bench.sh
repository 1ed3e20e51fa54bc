cd "${GRAFT_REPO_ROOT:?}"
T=${1:-bisect}
unset RTS_AMD_LIB
for rep in 1 2 3; do
  RTS_AMD_LIB=variants/librts_r02.so python3 bench.py --no-cpu-baseline --steps 64 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('r02', round(d['value']), round(d['ms_per_step'],3), {k:round(x,3) for k,x in d['config']['stage_ms_per_launch_rank0'].items()})" >> gpurun_out/${T}.log
  RTS_COOP_FRAC=0 python3 bench.py --no-cpu-baseline --steps 64 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('now, frac 0 (no k_tile_head)', round(d['value']), round(d['ms_per_step'],3), {k:round(x,3) for k,x in d['config']['stage_ms_per_launch_rank0'].items()})" >> gpurun_out/${T}.log
  python3 bench.py --no-cpu-baseline --steps 64 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('now', round(d['value']), round(d['ms_per_step'],3), {k:round(x,3) for k,x in d['config']['stage_ms_per_launch_rank0'].items()})" >> gpurun_out/${T}.log
done
cat gpurun_out/${T}.log
