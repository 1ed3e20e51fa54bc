#!/usr/bin/env python3
"""Static VALU opcode mix of the trace kernel's WALK LOOP (the depth-5 loop around the record fetch), priced with the measured
issue costs of profiles/<tag>_valu_calib.json.  tools/pmc_derive.py uses the result to price the VALU instructions that the
hardware counters do not classify (everything that is not f32 / f64 add, mul, fma or a transcendental): selects, min / max,
compares, integer and address arithmetic, moves.

  python tools/isa_mix.py <tag> [calibration tag]      writes profiles/<tag>_isa_mix.json   (needs hipcc; no GPU)
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL = "_Z7k_traceILb0ELb0ELb0ELb0ELb0ELb0ELb1EEv12RtsTraceArgs"      # k_trace<COUNT, KEEP_ALL, REFR, COOP, ASYNC, AFFINE, VERS>: the product kernel (octant versions walked)
# counted by SQ_INSTS_VALU_{ADD,MUL,FMA}_F32 / _F64 / TRANS_*: priced from the counters, not from this mix
COUNTED = re.compile(r"^v_(add|sub|subrev|mul|fma|fmac|mac|mad)_(f32|f64)$|^v_pk_(add|mul|fma)_f32$|^v_(rcp|rsq|sqrt|exp|log|sin|cos)_(f32|f64)$")


def opcode_costs(calib):
    c = {}
    for name, v in calib["classes"].items():
        c[name.split("_sgpr")[0].split("_to_sgpr")[0]] = v["waves_per_simd_4"]["cycles_per_wave_inst_per_simd"]
    c["v_cndmask_b32"] = calib["classes"]["v_cndmask_b32_sgpr_mask"]["waves_per_simd_4"]["cycles_per_wave_inst_per_simd"]   # (the VCC-chain figure of the probe does not occur in the kernel: DESIGN.md)
    return c


def price(op, costs, fast, slow):
    if op in costs:
        return costs[op], True
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base in costs:
        return costs[base], True
    fam = re.sub(r"_(i32|u32|b32|f32|i64|u64|b64|f64|u16|i16|f16)$", "", base)
    for k, v in costs.items():                      # same operation on another type of the same width class
        if re.sub(r"_(i32|u32|b32|f32)$", "", k) == fam and base.endswith(("i32", "u32", "b32", "f32")):
            return v, False
    return (slow if ("64" in base or base.startswith(("v_cmp", "v_min", "v_max", "v_med", "v_lsh", "v_ash", "v_bfe", "v_mad", "v_mul_", "v_cvt", "v_readlane", "v_writelane", "v_perm", "v_mbcnt", "v_div"))) else fast), False


def main():
    tag = sys.argv[1]; calib_tag = sys.argv[2] if len(sys.argv) > 2 else tag      # (the issue costs are the chip's: one calibration serves every build)
    calib = json.load(open(os.path.join(ROOT, "profiles", "%s_valu_calib.json" % calib_tag)))
    costs = opcode_costs(calib)
    fast = costs["v_add_u32"]; slow = costs["v_min_f32"]
    src = os.path.join(ROOT, "rts_amd", "csrc", "rts_trace.hip")
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "t.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                               "-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", src, "-o", out], stderr=subprocess.DEVNULL)
        isa = open(out).read()
    body = isa[isa.index("\n" + KERNEL + ":"):]
    lines = body[:body.index("s_endpgm")].splitlines()
    first = [i for i, l in enumerate(lines) if "global_load_dwordx4" in l and "offset" not in l][0]
    hdr = None
    for j in range(first, 0, -1):
        m = re.search(r"Header=(BB\d+_\d+) Depth=(\d+)", lines[j])
        if m:
            hdr = m.group(1); break
    hist = collections.Counter(); inloop = False
    for l in lines:
        t = l.strip()
        if t.startswith(".LBB") or t.startswith("; %bb"):
            inloop = ("Header=%s " % hdr) in t or t.startswith(".L" + hdr + ":"); continue
        m = re.match(r"(v_\w+)", t)
        if inloop and m:
            hist[re.sub(r"_(e32|e64)$", "", m.group(1))] += 1
    other = {op: n for op, n in hist.items() if not COUNTED.match(op)}
    n_other = sum(other.values()); cyc = 0.0; detail = {}; unmeasured = 0
    for op, n in sorted(other.items(), key=lambda kv: -kv[1]):
        c, measured = price(op, costs, fast, slow)
        cyc += c * n; detail[op] = dict(count=n, cycles=c, measured=measured); unmeasured += 0 if measured else n
    res = dict(tag=tag, calibration="profiles/%s_valu_calib.json" % calib_tag, kernel=KERNEL, loop_header=hdr, valu_in_loop=sum(hist.values()), uncounted_valu_in_loop=n_other,
               cycles_per_uncounted_valu=cyc / max(n_other, 1), unmeasured_share=unmeasured / max(n_other, 1),
               fast_class_cycles=fast, slow_class_cycles=slow,
               note="static mix of the walk loop (every block of the loop counted once); opcodes without a calibration run are priced by their family (64-bit, compare, min/max, shift, convert, lane ops: slow class; the rest: fast class)",
               opcodes=detail, counted_classes={op: n for op, n in hist.items() if COUNTED.match(op)})
    dst = os.path.join(ROOT, "profiles", "%s_isa_mix.json" % tag)
    json.dump(res, open(dst, "w"), indent=1, sort_keys=True)
    print(dst, "uncounted VALU in the loop:", n_other, "at", round(res["cycles_per_uncounted_valu"], 3), "cycles; unmeasured share", round(res["unmeasured_share"], 3))


if __name__ == "__main__":
    main()
