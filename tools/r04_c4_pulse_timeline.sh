cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out; rm -rf gpurun_out/prof_c4tl
R=$GRAFT_REPO_ROOT
RTS_DEBUG_COOP=1 python bench.py --config c4 --steps 12 --warmup 12 --no-cpu-baseline > gpurun_out/r04x_c4_dbg.json 2> gpurun_out/r04x_c4_dbg.err
grep "head hint" gpurun_out/r04x_c4_dbg.err | awk '{print $8, $11}' | tr '\n' ';' | cut -c1-600; echo
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_c4tl -- python3 $R/bench.py --config c4 --steps 12 --warmup 12 --no-cpu-baseline > $R/gpurun_out/r04x_c4_under_rocprof.json 2>/dev/null
cd $R && python - <<'PY' > gpurun_out/r04x_c4_pulse_timeline.log
import csv, glob
f = sorted(glob.glob("gpurun_out/prof_c4tl/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
ev = sorted([(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:60], r.get("Queue_Id", "?"), r.get("Grid_Size","?")) for r in rows])
tr = [e for e in ev if e[2].startswith("k_trace")]
t0 = tr[0][0]
for s, e, k, q, g in tr:
    print("%9.3f -> %9.3f  (%7.3f ms)  q%-3s grid %-8s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, (e - s) / 1e6, q, g, "COOP" if "false, true, false, false>" in k else k[:40]))
PY
find gpurun_out/prof_c4tl -name "*.csv" -delete
python tools/bench_line.py gpurun_out/r04x_c4_under_rocprof.json
