#!/usr/bin/env python3
"""Turns the rocprofv3 --pmc passes of tools/pmc_collect.sh into the per-launch figures bench.py reports.

  python tools/pmc_derive.py <tag> <workload>     reads gpurun_out/pmc_<tag>_<workload>/<pass>/**/_counter_collection.csv
                                                   writes profiles/<tag>_pmc_<workload>.json and one trimmed CSV per pass
                                                   (profiles/<tag>_pmc_<workload>_<pass>.csv: the k_trace dispatches only)

Every utilisation is formed INSIDE one pass -- busy cycles of a unit over (instances x GRBM_GUI_ACTIVE / 8 XCDs of the
same dispatches) -- so it cannot exceed 1 and does not mix a profiled duration with an un-profiled one.  The first
dispatch of a pass (first launch of the handle: index-order tiles, cold caches) is dropped.
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_XCD, N_CU, N_SIMD = 8, 256, 1024
PEAK_CLOCK_HZ = 2.4e9                      # MI355X_MICROARCH.md: max clock


def kernel_rows(path):
    """-> (ordinary launches, cooperative launches, meta): per dispatch {counter: value, "_ns": duration}.  A launch of the
    library is the ordinary trace kernel k_trace<.., COOP = false, ..> and, when the handle's cost order has a head, the cooperative
    kernel k_trace<.., COOP = true, ..> beside it (rts_trace.hip); under counter collection rocprofv3 runs the dispatches one at a time."""
    rows = {False: collections.OrderedDict(), True: collections.OrderedDict()}
    meta = {}
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "k_trace" not in name:
            continue
        targs = [x.strip() for x in name.split("(")[0].split("<", 1)[-1].rstrip("> ").split(",")]      # k_trace<COUNT, KEEP_ALL, REFR, COOP[, ASYNC]>
        coop = len(targs) >= 4 and targs[3] == "true"
        d = rows[coop].setdefault(int(r["Dispatch_Id"]), {})
        d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        d["_ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        if not coop:
            meta = dict(kernel=name, vgpr=int(r["VGPR_Count"]), agpr=int(r["Accum_VGPR_Count"]), sgpr=int(r["SGPR_Count"]),
                        lds=int(r["LDS_Block_Size"]), scratch=int(r["Scratch_Size"]), grid=int(r["Grid_Size"]), wg=int(r["Workgroup_Size"]))
    return list(rows[False].values()), list(rows[True].values()), meta


def main():
    tag, wl = sys.argv[1], sys.argv[2]
    src = os.path.join(ROOT, "gpurun_out", "pmc_%s_%s" % (tag, wl))
    out = dict(tag=tag, workload=wl, source="tools/pmc_collect.sh %s %s (rocprofv3 --pmc, one pass per counter set; program: python3 tools/trace_bench.py %s)" % (tag, wl, wl),
               passes={}, per_launch={})
    meta = {}
    for pdir in sorted(glob.glob(os.path.join(src, "*/"))):
        name = os.path.basename(pdir.rstrip("/"))
        files = glob.glob(os.path.join(pdir, "**", "*_counter_collection.csv"), recursive=True)
        if not files:
            continue
        files.sort(key=os.path.getmtime, reverse=True)          # (a re-collected tag: gpurun merges new files beside the old ones)
        rows, crows, m = kernel_rows(files[0])
        if len(rows) < 2:
            continue
        meta = m or meta
        use = rows[1:]                                                      # (first launch of the handle: index-order tiles, cold caches)
        if len(crows) >= 2:                                                 # launches with a cooperative kernel: the steady state starts AFTER the first of them (it runs on a hint one launch old)
            k = len(crows) - 1
            use = rows[-k:]; crows = crows[-k:]
        avg = {k: sum(r.get(k, 0.0) for r in use) / len(use) for k in use[0]}
        ordinary_us = avg["_ns"] / 1e3
        coop_us = 0.0
        if crows:                                                           # the cooperative kernel of a launch: counters ADD, and so do the durations (dispatches are serialised under --pmc)
            cavg = {k: sum(r.get(k, 0.0) for r in crows) / len(crows) for k in crows[0]}
            coop_us = cavg["_ns"] / 1e3
            out.setdefault("coop_per_launch", {}).update({k: v for k, v in cavg.items() if k != "_ns"})
            for k, v in cavg.items():
                avg[k] = avg.get(k, 0.0) + v
        out["passes"][name] = dict(dispatches=len(use), coop_dispatches=len(crows), kernel_us=avg.pop("_ns") / 1e3, ordinary_kernel_us=ordinary_us, coop_kernel_us=coop_us, counters=avg)
        out["per_launch"].update(avg)
        # trimmed copy of the pass for profiles/ (k_trace dispatches only)
        with open(files[0]) as fi, open(os.path.join(ROOT, "profiles", "%s_pmc_%s_%s.csv" % (tag, wl, name)), "w") as fo:
            for i, line in enumerate(fi):
                if i == 0 or "k_trace" in line:
                    fo.write(line)
    un = os.path.join(src, "unprofiled.log")
    if os.path.exists(un):
        out["unprofiled_line"] = open(un).read().strip().splitlines()[-1]
        import re
        m = re.search(r"segs (\d+)", out["unprofiled_line"])
        if m:
            out["segments_per_launch"] = int(m.group(1))
    P = out["passes"]; L = out["per_launch"]
    d = {}

    def cyc(name):                           # kernel cycles per XCD of the pass that carries `name`
        for p in P.values():
            if name in p["counters"] and "GRBM_GUI_ACTIVE" in p["counters"]:
                return p["counters"]["GRBM_GUI_ACTIVE"] / N_XCD
        return None
    if "SQ_ACTIVE_INST_VALU" in L and cyc("SQ_ACTIVE_INST_VALU"):
        k = cyc("SQ_ACTIVE_INST_VALU")
        d["kernel_cycles"] = k
        d["valu_busy"] = L["SQ_ACTIVE_INST_VALU"] * 4.0 / (N_SIMD * k)          # SQ_ACTIVE_INST_* count quad-cycles per wave
        d["scalar_busy"] = L["SQ_ACTIVE_INST_SCA"] * 4.0 / (N_SIMD * k)
        d["active_lanes_per_valu_inst"] = L["SQ_THREAD_CYCLES_VALU"] / max(L["SQ_ACTIVE_INST_VALU"], 1.0) if "SQ_THREAD_CYCLES_VALU" in L else None
        d["waves_per_simd_avg"] = L["SQ_WAVE_CYCLES"] * 4.0 / (N_SIMD * k) if "SQ_WAVE_CYCLES" in L else None
    if "SQ_INSTS_VALU" in L:
        d["valu_wave_insts"] = L["SQ_INSTS_VALU"]
        d["vmem_rd_wave_insts"] = L.get("SQ_INSTS_VMEM_RD"); d["vmem_wr_wave_insts"] = L.get("SQ_INSTS_VMEM_WR")
        d["salu_wave_insts"] = L.get("SQ_INSTS_SALU"); d["lds_wave_insts"] = L.get("SQ_INSTS_LDS"); d["waves"] = L.get("SQ_WAVES")
        # 4 cycles of one SIMD per VALU wave-instruction (f32 and f64 alike on gfx950: FP64 vector = FP32 non-packed rate)
        if "kernel_cycles" in d:
            d["valu_issue_frac_from_counts"] = L["SQ_INSTS_VALU"] * 4.0 / (N_SIMD * d["kernel_cycles"])
            # vector-memory return path, from COUNTS: a wave-wide dwordx4 load returns 64 x 16 B = 1024 B through the CU's
            # 64 B/clk L1 -> register path = 16 clk whatever the addresses; nearly every read of this kernel is one (the
            # record fetch); priced at 16 clk each this is an upper bound of the data cycles
            d["vmem_return_frac_from_counts"] = L.get("SQ_INSTS_VMEM_RD", 0.0) * 16.0 / (N_CU * d["kernel_cycles"])
    if "SQ_INSTS_VALU_ADD_F64" in L:
        d["f64_wave_insts"] = sum(L.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
    # ---- CALIBRATED VALU issue demand.  A wave-instruction does not cost "4 cycles": measured on this
    # chip with >= 2 waves per SIMD (tools/valu_calib.hip -> profiles/<calib>_valu_calib.json) f32 add / mul / fma and simple
    # integer / logic / move instructions cost 2.25 cycles of their SIMD, f64 arithmetic, min / max, compares, selects, shifts
    # and conversions 4.1-4.3, v_rcp_f32 8.1, v_rcp_f64 16.1.  The counters classify f32 / f64 add, mul, fma and the
    # transcendentals; the rest is priced with the static opcode mix of the walk loop (tools/isa_mix.py).
    calib_tag = os.environ.get("RTS_CALIB_TAG", "r03")
    cp, mp = os.path.join(ROOT, "profiles", "%s_valu_calib.json" % calib_tag), os.path.join(ROOT, "profiles", "%s_isa_mix.json" % calib_tag)
    if os.path.exists(os.path.join(ROOT, "profiles", "%s_isa_mix.json" % tag)):      # the opcode mix of THIS build's walk loop (tools/isa_mix.py <tag> <calibration tag>)
        mp = os.path.join(ROOT, "profiles", "%s_isa_mix.json" % tag)
    if "SQ_INSTS_VALU_ADD_F64" in L and "SQ_INSTS_VALU" in L and os.path.exists(cp) and os.path.exists(mp) and "kernel_cycles" in d:
        cal = json.load(open(cp))["classes"]; mix = json.load(open(mp))
        c = lambda k: cal[k]["waves_per_simd_4"]["cycles_per_wave_inst_per_simd"]
        f32 = sum(L.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F32", "SQ_INSTS_VALU_MUL_F32", "SQ_INSTS_VALU_FMA_F32"))
        f64 = sum(L.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64"))
        t32, t64 = L.get("SQ_INSTS_VALU_TRANS_F32", 0.0), L.get("SQ_INSTS_VALU_TRANS_F64", 0.0)
        other = max(L["SQ_INSTS_VALU"] - f32 - f64 - t32 - t64, 0.0)
        vcyc = f32 * c("v_fma_f32") + f64 * c("v_fma_f64") + t32 * c("v_rcp_f32") + t64 * c("v_rcp_f64") + other * mix["cycles_per_uncounted_valu"]
        d["valu_classes"] = dict(f32_add_mul_fma=f32, f64_add_mul_fma=f64, trans_f32=t32, trans_f64=t64, other=other,
                                 cycles_each=dict(f32=c("v_fma_f32"), f64=c("v_fma_f64"), trans_f32=c("v_rcp_f32"), trans_f64=c("v_rcp_f64"), other=mix["cycles_per_uncounted_valu"]),
                                 source=[os.path.relpath(cp, ROOT), os.path.relpath(mp, ROOT)])
        d["valu_issue_cycles_calibrated"] = vcyc
        d["valu_cycles_per_inst_calibrated"] = vcyc / L["SQ_INSTS_VALU"]
        d["valu_issue_frac_calibrated"] = vcyc / (N_SIMD * d["kernel_cycles"])
    if "TA_TA_BUSY_sum" in L and cyc("TA_TA_BUSY_sum"):
        d["ta_busy"] = L["TA_TA_BUSY_sum"] / (N_CU * cyc("TA_TA_BUSY_sum"))
    if "TD_TD_BUSY_sum" in L and cyc("TD_TD_BUSY_sum"):
        d["td_busy"] = L["TD_TD_BUSY_sum"] / (N_CU * cyc("TD_TD_BUSY_sum"))
        d["td_busy_cycles"] = L["TD_TD_BUSY_sum"]
        # NOT a utilisation of the return path: TD_TD_BUSY counts cycles with a request in the unit -- an (almost) empty launch
        # (tools/trace_bench.py c3empty: 1.4 M vector-memory instructions, 0.38 ms) shows 0.85 -- use vmem_return_frac_from_counts
    if "TCP_TOTAL_CACHE_ACCESSES_sum" in L:
        d["l1_line_accesses"] = L["TCP_TOTAL_CACHE_ACCESSES_sum"]
        if "TCP_TCC_READ_REQ_sum" in L:
            d["l1_hit_rate"] = 1.0 - L["TCP_TCC_READ_REQ_sum"] / max(L["TCP_TOTAL_CACHE_ACCESSES_sum"], 1.0)
        if d.get("vmem_rd_wave_insts"):
            d["l1_lines_per_vmem_inst"] = L["TCP_TOTAL_CACHE_ACCESSES_sum"] / d["vmem_rd_wave_insts"]
    if "TCC_HIT_sum" in L:
        d["l2_hit_rate"] = L["TCC_HIT_sum"] / max(L["TCC_HIT_sum"] + L["TCC_MISS_sum"], 1.0)
    if "SQ_WAIT_ANY" in L and "SQ_WAVE_CYCLES" in L:
        d["wave_wait_any_frac"] = L["SQ_WAIT_ANY"] / L["SQ_WAVE_CYCLES"]
        d["wave_wait_inst_frac"] = L["SQ_WAIT_INST_ANY"] / L["SQ_WAVE_CYCLES"]
    if "FETCH_SIZE" in L and "WRITE_SIZE" in L:
        # FETCH_SIZE / WRITE_SIZE are in KiB; gfx950 tallies 128-B read requests at 64 B (MI355X_MICROARCH.md, HBM): the x2
        # correction is calibrated for wide streaming reads, for this kernel's divergent reads it is an upper bound
        d["hbm_fetch_bytes_raw"] = L["FETCH_SIZE"] * 1024.0; d["hbm_fetch_bytes_x2"] = 2.0 * L["FETCH_SIZE"] * 1024.0
        d["hbm_write_bytes"] = L["WRITE_SIZE"] * 1024.0
        d["hbm_bytes_per_launch"] = d["hbm_fetch_bytes_x2"] + d["hbm_write_bytes"]
    out["derived"] = d
    out["kernel_resources"] = meta
    # the sources the profiled library was built from (rts_build_id): bench.py prices a run with these counters only if the
    # library it loaded carries the same hash
    sys.path.insert(0, ROOT)
    try:
        from rts_amd import _lib
        out["source_hash"] = os.environ.get("RTS_PROFILE_HASH") or _lib.source_hash()
    except Exception as e:                                                   # pragma: no cover
        out["source_hash"] = None; out["source_hash_error"] = str(e)
    dst = os.path.join(ROOT, "profiles", "%s_pmc_%s.json" % (tag, wl))
    json.dump(out, open(dst, "w"), indent=1, sort_keys=True)
    print(dst)
    for k in sorted(d):
        print("  %-32s %s" % (k, d[k]))
    print("  resources", meta)


if __name__ == "__main__":
    main()
