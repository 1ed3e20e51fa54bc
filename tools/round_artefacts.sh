# the judged artefacts of a round, in two calls (the bench lines price themselves with counters of the SAME build):
#   tools/round_artefacts.sh <tag> pmc      counters of c3 / dense control / c2 / c4  -> then, on the CPU: tools/pmc_derive.py <tag> <workload>, commit
#   tools/round_artefacts.sh <tag> bench    bench lines, kernel stats, timelines, CPU baseline, adapter bench, RCCL world-1 line
cd "${GRAFT_REPO_ROOT:?}"
T=${1:-r04}; WHAT=${2:-bench}
mkdir -p gpurun_out
if [ "$WHAT" = pmc ]; then
  for w in c3 c3narrow c2 c4 c5; do bash tools/pmc_collect.sh $T $w 6 > gpurun_out/${T}_pmc_$w.log 2>&1; tail -1 gpurun_out/${T}_pmc_$w.log; done
  exit 0
fi
python bench.py --steps 20 --warmup 5 > gpurun_out/${T}_bench_c3_steps20.json 2> gpurun_out/${T}_bench_c3_steps20.err
python bench.py --no-cpu-baseline > gpurun_out/${T}_bench_c3_default_256steps.json 2>/dev/null
python bench.py --no-cpu-baseline --inflight 1 --steps 64 > gpurun_out/${T}_bench_c3_inflight1.json 2>/dev/null
python bench.py --no-cpu-baseline --config c3ecef --steps 64 > gpurun_out/${T}_bench_c3ecef.json 2>/dev/null
python bench.py --no-cpu-baseline --config c2 --steps 64 > gpurun_out/${T}_bench_c2.json 2>/dev/null
python bench.py --no-cpu-baseline --config c4 --steps 24 --warmup 12 > gpurun_out/${T}_bench_c4.json 2>/dev/null
python bench.py --no-cpu-baseline --config c4 --tx both --steps 24 --warmup 12 > gpurun_out/${T}_bench_c4_2tx.json 2>/dev/null
python bench.py --no-cpu-baseline --config c5 --steps 1024 --warmup 16 > gpurun_out/${T}_bench_c5.json 2>/dev/null
python bench.py --no-cpu-baseline --config sphere6 --steps 256 > gpurun_out/${T}_bench_sphere6.json 2>/dev/null
python bench.py --no-cpu-baseline --steps 64 --fused-post > gpurun_out/${T}_bench_c3_fused_post.json 2>/dev/null
RTS_BUILDER=host python bench.py --no-cpu-baseline --steps 64 > gpurun_out/${T}_bench_c3_host_tree.json 2>/dev/null
for f in c3_steps20 c3_default_256steps c3_inflight1 c3ecef c2 c4 c4_2tx c5 sphere6 c3_fused_post c3_host_tree; do python -c "
import json
j=json.loads(open('gpurun_out/${T}_bench_$f.json').read().strip().splitlines()[-1]); r=j['roofline']
print('$f', round(j['value']), 'Mrays/s', round(j['ms_per_step'],3), 'ms/pulse | serial', round(r['kernel_ms_serial'],3), 'hit', round(r['hit_fraction'],3), '| bound', r['bound'], r['frac'], r.get('frac_if_every_inst_cost_4_cycles'), 'setup_s', round(j['config']['scene_setup_s'],3))"; done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$T -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${T}_bench_under_rocprof.json 2> /dev/null; cd $GRAFT_REPO_ROOT
find gpurun_out/prof_$T -name "*_kernel_trace.csv" -delete; find gpurun_out/prof_$T -name "*_agent_info.csv" -delete
python tools/timeline.py c3 > gpurun_out/${T}_timeline_c3.log 2>&1; python tools/timeline.py c4 > gpurun_out/${T}_timeline_c4.log 2>&1; rm -f gpurun_out/timeline.bin gpurun_out/tile_us.npy
python tools/count_stats.py c3 c3narrow c2 c4 > gpurun_out/${T}_count_stats.log 2>&1
tools/adapter_bench_bin 216 256 3 6 6 5 dh > gpurun_out/${T}_adapter_bench.json 2> gpurun_out/${T}_adapter_bench.err; python tools/adapter_line.py gpurun_out/${T}_adapter_bench.json
tools/adapter_bench_bin 216 256 1 6 6 3 d > gpurun_out/${T}_adapter_bench_inflight1.json 2>/dev/null; python tools/adapter_line.py gpurun_out/${T}_adapter_bench_inflight1.json
for r in 0 1 2 3 4 5 6 7; do echo "part $r/8: $(RTS_SHARD=8 RTS_SHARD_PART=$r RTS_VERBOSE=1 python tools/trace_bench.py c4 8 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-300)" >> gpurun_out/${T}_c4_eighths_final.log; done
python tools/deal_bench.py c4 8 4096 > gpurun_out/${T}_c4_deal.log 2>&1; tail -2 gpurun_out/${T}_c4_deal.log
echo "whole: $(RTS_VERBOSE=1 python tools/trace_bench.py c4 10 2>&1 | tail -2 | tr '\n' ' ' | cut -c1-330)" >> gpurun_out/${T}_c4_eighths_final.log
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --backend nccl > gpurun_out/${T}_bench_rccl_world1.json 2> gpurun_out/${T}_bench_rccl_world1.err
python tools/cpu_baseline.py c1 c2 c3 > gpurun_out/${T}_cpu_baseline.log 2>&1; cat gpurun_out/${T}_cpu_baseline.log
python tools/scene_info.py c3 c4 > gpurun_out/${T}_scene_info.log 2>&1; cat gpurun_out/${T}_scene_info.log
