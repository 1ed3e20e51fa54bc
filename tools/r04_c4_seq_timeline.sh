cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out; rm -rf gpurun_out/prof_c4s; R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/prof_c4s -- python3 $R/bench.py --config c4 --inflight 1 --steps 8 --warmup 6 --no-cpu-baseline > $R/gpurun_out/r04x_c4s.json 2>/dev/null )
python tools/bench_line.py gpurun_out/r04x_c4s.json | cut -c1-200
python - <<'PY' > gpurun_out/r04_c4_inflight1_pulse_timeline.log
import csv, glob
f = sorted(glob.glob("gpurun_out/prof_c4s/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:60], "q" + r.get("Queue_Id", "?")) for r in rows]
for g in glob.glob("gpurun_out/prof_c4s/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(g)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Name", "?"))[:40], ""))
ev.sort()
tr = [i for i, e in enumerate(ev) if e[2].startswith("k_trace<false, false, false, false, false")]
i0 = tr[9]; i1 = tr[10]
t0 = ev[i0][0]
print("one sequential C4 pulse of bench.py --config c4 --inflight 1 (us from its ordinary trace kernel's start; start -> end (duration) [gap before])")
prev_end = None
for s, e, k, q in ev[i0 - 8:i1 + 1]:
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%9.1f -> %9.1f (%7.1f) [gap %6.1f] %-4s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, gap, q, k)); prev_end = max(e, prev_end or e)
PY
find gpurun_out/prof_c4s -name "*.csv" -delete
cat gpurun_out/r04_c4_inflight1_pulse_timeline.log | cut -c1-150
