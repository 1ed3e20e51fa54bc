# C4 (BASELINE configs[3]) bench under a kernel trace for a list of environment settings: pulse rate, and the ordinary / cooperative
# kernel durations of the serial launches at the end of the bench
cd "${GRAFT_REPO_ROOT:?}"; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
run() {   # name, env...
  name=$1; shift
  rm -rf gpurun_out/prof_scan
  ( cd /tmp && export TMPDIR=/tmp && env "$@" rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_scan -- python3 $R/bench.py --config c4 --steps 12 --warmup 12 --no-cpu-baseline > $R/gpurun_out/scan_$name.json 2>/dev/null )
  python - "$name" <<'PY'
import csv, glob, json, sys
name = sys.argv[1]
f = sorted(glob.glob("gpurun_out/prof_scan/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
ev = sorted([(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows if r["Kernel_Name"].startswith("void k_trace<false")])
co = [e for e in ev if "false, true, false, false>" in e[2]]; od = [e for e in ev if "false, true, false, false>" not in e[2]]
# serial phase: the last 10 ordinary launches before the two closing ones (counting build, keep-all) -- take launches -12..-2
ser_o = od[-12:-2]; t_lo = ser_o[0][0]
ser_c = [e for e in co if e[0] >= t_lo - 1000000]
pip_o = od[6:-12]; pip_c = [e for e in co if e[0] < t_lo - 1000000]
j = json.loads(open("gpurun_out/scan_%s.json" % name).read().strip().splitlines()[-1])
m = lambda v: sum((e[1] - e[0]) for e in v) / max(1, len(v)) / 1e6
print("%-26s %7.3f ms/pulse | serial launches: ordinary %6.3f coop %6.3f ms (n %d / %d) | pipelined: ordinary %6.3f coop %6.3f ms (n %d / %d)" % (name, j["ms_per_step"], m(ser_o), m(ser_c), len(ser_o), len(ser_c), m(pip_o), m(pip_c), len(pip_o), len(pip_c)))
PY
  find gpurun_out/prof_scan -name "*.csv" -delete
}
run default A=1
run hwq8 GPU_MAX_HW_QUEUES=8
run hwq12 GPU_MAX_HW_QUEUES=12
run frac0.75 RTS_COOP_FRAC=0.75
run frac1.0 RTS_COOP_FRAC=1.0
run frac1.5 RTS_COOP_FRAC=1.5
run frac0.35 RTS_COOP_FRAC=0.35

