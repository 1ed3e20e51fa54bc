#!/bin/bash
# Collects the hardware counters that say what bounds k_trace, one rocprofv3 pass per counter set (each set fits the
# per-block slot limits of gfx950: SQ 8, TCC 4 -- FETCH_SIZE costs 3, WRITE_SIZE 2 --, TA/TD/TCP 2-4, GRBM 2).
#   tools/pmc_collect.sh <tag> <workload> [reps]      e.g.  tools/pmc_collect.sh r02a c3 6
# Writes gpurun_out/pmc_<tag>_<workload>/<pass>/ ; tools/pmc_derive.py turns them into profiles/<tag>_pmc_<workload>.json.
# The program stands directly after `--` (no env/bash hop: the profiler's preload has already initialised the GPU) and
# tools/trace_bench.py never builds anything (a stale librts_amd.so is an error, not a make).
set -u
TAG=$1; WL=$2; REPS=${3:-6}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/pmc_${TAG}_${WL}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
declare -A PASS
PASS[insts]="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_FLAT"
PASS[f64]="SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32"
PASS[active]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
PASS[wait]="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SMEM SQ_INST_CYCLES_SALU SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
PASS[ta]="TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum GRBM_GUI_ACTIVE GRBM_TA_BUSY"
PASS[ta2]="TA_FLAT_READ_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE"
PASS[td]="TD_TD_BUSY_sum TD_LOAD_WAVEFRONT_sum GRBM_GUI_ACTIVE"
PASS[tcp]="TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum GRBM_GUI_ACTIVE"
PASS[tcp2]="TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE"
PASS[tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
PASS[fetch]="FETCH_SIZE"
PASS[write]="WRITE_SIZE"
rc_all=0
for p in insts f64 active wait ta ta2 td tcp tcp2 tcc fetch write; do
    echo "== pass $p: ${PASS[$p]}"
    timeout -k 10 240 rocprofv3 --pmc ${PASS[$p]} --kernel-trace --output-format csv -d "$OUT/$p" -- python3 "$ROOT/tools/trace_bench.py" "$WL" "$REPS" > "$OUT/$p.log" 2>&1
    rc=$?
    tail -1 "$OUT/$p.log"
    find "$OUT/$p" -name "*_kernel_trace.csv" -delete 2>/dev/null; find "$OUT/$p" -name "*_agent_info.csv" -delete 2>/dev/null      # (gpurun copies back 64 MiB at most: the counter tables are what tools/pmc_derive.py reads)
    for f in $(find "$OUT/$p" -name "*_counter_collection.csv"); do { head -1 "$f"; grep k_trace "$f"; } > "$f.tmp" && mv "$f.tmp" "$f"; done      # ... and of them the trace kernels' rows (the device hierarchy build alone is ~1 000 dispatches)
    if [ $rc -ne 0 ]; then echo "pass $p failed rc=$rc"; rc_all=1; if [ $rc -ge 124 ]; then echo "timeout/kill: stopping"; exit 1; fi; fi
done
# un-profiled reference timing of the same command
python3 "$ROOT/tools/trace_bench.py" "$WL" "$REPS" > "$OUT/unprofiled.log" 2>&1; tail -1 "$OUT/unprofiled.log"
exit $rc_all
