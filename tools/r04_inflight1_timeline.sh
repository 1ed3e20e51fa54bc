cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out; rm -rf gpurun_out/prof_if1; R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/prof_if1 -- python3 $R/bench.py --inflight 1 --steps 40 --warmup 8 --no-cpu-baseline > $R/gpurun_out/r04x_if1.json 2>/dev/null )
python tools/bench_line.py gpurun_out/r04x_if1.json | cut -c1-200
python - <<'PY' > gpurun_out/r04_inflight1_pulse_timeline.log
import csv, glob
f = sorted(glob.glob("gpurun_out/prof_if1/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")[:48], "q" + r.get("Queue_Id", "?")) for r in rows]
for g in glob.glob("gpurun_out/prof_if1/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(g)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", r.get("Name", "?"))[:40], ""))
ev.sort()
tr = [i for i, e in enumerate(ev) if e[2].startswith("k_trace<false, false, false, false")]
i0 = tr[20]; i1 = tr[23]          # three pulses of the timed interval
t0 = ev[i0][0]
print("three sequential pulses of bench.py --inflight 1 (us from the first trace kernel's start; start -> end (duration) [gap before])")
prev_end = None
for s, e, k, q in ev[i0 - 6:i1 + 1]:
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print("%9.1f -> %9.1f (%7.1f) [gap %6.1f] %-4s %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, gap, q, k)); prev_end = max(e, prev_end or e)
PY
find gpurun_out/prof_if1 -name "*.csv" -delete
cat gpurun_out/r04_inflight1_pulse_timeline.log | cut -c1-150
