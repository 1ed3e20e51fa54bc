cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
RTS_DEBUG_COOP=1 python bench.py --config c4 --steps 12 --warmup 12 --no-cpu-baseline > gpurun_out/r04x_c4_dbg.json 2> gpurun_out/r04x_c4_dbg.err
echo "C4 after the guard: head counts $(grep 'end:' gpurun_out/r04x_c4_dbg.err | awk '{print $7}' | tr '\n' ' ')"; python tools/bench_line.py gpurun_out/r04x_c4_dbg.json
python bench.py --config c4 --steps 24 --warmup 12 --no-cpu-baseline > gpurun_out/r04x_c4.json 2>/dev/null; python tools/bench_line.py gpurun_out/r04x_c4.json
for q in 4 8; do
rm -rf gpurun_out/prof_q
( cd /tmp && export TMPDIR=/tmp && GPU_MAX_HW_QUEUES=$q rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_q -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r04x_c3_q$q.json 2>/dev/null )
python tools/bench_line.py gpurun_out/r04x_c3_q$q.json
python - <<'PY'
import csv, glob, collections
f = sorted(glob.glob("gpurun_out/prof_q/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
ev = sorted([(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0].replace("void ", ""), r["Queue_Id"]) for r in rows])
tr = [e for e in ev if e[2] == "k_trace"]
win = tr[10:34]; t0 = win[0][0]
print("queues of 24 pipelined trace launches:", " ".join(e[3] for e in win))
byq = collections.defaultdict(set)
for s, e, k, q in ev:
    if win[0][0] <= s <= win[-1][1]: byq[q].add(k)
for q in sorted(byq): print("  queue", q, sorted(byq[q]))
# overlap: how many trace kernels are resident on average over the window
import numpy as np
ts = sorted([(e[0], 1) for e in win] + [(e[1], -1) for e in win]); cur = 0; last = ts[0][0]; acc = collections.Counter()
for t, d in ts: acc[cur] += t - last; last = t; cur += d
tot = sum(acc.values()); print("  time share with n trace kernels resident:", {k: round(v / tot, 3) for k, v in sorted(acc.items())}, " window %.3f ms for 24 launches = %.4f ms each" % (tot / 1e6, tot / 24e6))
PY
find gpurun_out/prof_q -name "*.csv" -delete
done
