cd $GRAFT_REPO_ROOT
python3 bench.py --no-cpu-baseline --steps 64 --link > gpurun_out/r02d_bench_c3_link.json 2>/dev/null
python3 bench.py --no-cpu-baseline --steps 64 --inflight 1 > gpurun_out/r02d_bench_c3_inflight1.json 2>/dev/null
python3 bench.py --no-cpu-baseline --steps 64 --inflight 2 > gpurun_out/r02d_bench_c3_inflight2.json 2>/dev/null
for f in link inflight1 inflight2; do python3 -c "
import json
j=json.loads(open('gpurun_out/r02d_bench_c3_$f.json').read().strip().splitlines()[-1]); r=j['roofline']
print('$f', round(j['value']), round(j['ms_per_step'],3), 'serial', round(r['kernel_ms_serial'],3), j['config']['stage_ms_per_launch_rank0'])"; done
echo "device builder c3: $(RTS_BUILDER=device python3 tools/trace_bench.py c3 12 | tail -1)"
echo "host builder c3: $(python3 tools/trace_bench.py c3 12 | tail -1)"
echo "device builder dense: $(RTS_BUILDER=device python3 tools/trace_bench.py c3narrow 6 | tail -1)"
RTS_BUILDER=device python3 tools/count_stats.py 2>/dev/null | grep -E "^c3 "
python3 tools/count_stats.py 2>/dev/null
python3 - <<'PY'
import sys, os, time
sys.path.insert(0, os.getcwd())
from rts_amd import api, scenes
for name, spec in (("c3", scenes.config3()), ("c4", scenes.config4())):
    for dev in (False, True):
        tr = api.Tracer(spec["W"], spec["max_refl"], 0, spec["smooth"], device_build=dev)
        t0 = time.time(); tr.set_scene(spec["meshes"]); dt = time.time() - t0
        tr2 = api.Tracer(spec["W"], spec["max_refl"], 0, spec["smooth"]); t1 = time.time(); tr2.share_scene(tr); ds = time.time() - t1
        print(name, "device" if dev else "host", "set_scene %.1f ms, share %.2f ms" % (dt * 1e3, ds * 1e3), tr.scene_info())
        tr2.close(); tr.close()
PY
