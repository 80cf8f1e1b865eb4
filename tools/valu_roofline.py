#!/usr/bin/env python3
"""Vector-issue roofline of the sector-query kernel.

The kernel is bound by the rate at which the SIMDs can issue its vector instructions, not by HBM.  Its ceiling is
therefore: (vector wave-instructions per launch, each priced with its measured issue cost in SIMD cycles) divided by
(SIMDs x shader clock).  This tool prices the instruction mix:

  * dynamic counts per launch from rocprofv3 counters (tools/pmc_mix.sh -> tools/pmc_summary.py): SQ_INSTS_VALU and
    its per-type sub-counters (f32 / f64 add, mul, fma; transcendental; int32; int64; conversions);
  * issue cost per opcode from tools/valu_issue_bench.hip (measured on the same GPU);
  * for the two buckets whose opcodes do not share one cost (INT32 and everything no sub-counter covers: moves,
    selects, compares, lane operations) the static opcode mix of the kernel's own code (hipcc -S) gives the average.

It also reports the hardware's own figure, SQ_ACTIVE_INST_VALU (in quad-cycles), which the priced sum should
reproduce.

    python tools/valu_roofline.py <pmc_summary.json> <issue_cost.json> <kernel.s> <symbol-substring> <out.json> [<name-in-counters>]
        [--sources file ...] [--workload JSON] [--work particles,cells_in_range,beams] [--counts counts.json]

--sources: the files the kernel is built from; their sha256 goes into the output, and bench.py refuses to quote a
profile whose sources are not the ones in the tree (`stale`).  --workload: what the counter passes were run on
(bench.py quotes a profile only for that workload).  --work: sizes for the ALGORITHMIC floor, below.

Two figures beside the issue fraction say how much of the issued work is needed at all:
  * lane utilisation: of the 64 lanes of a vector instruction, how many were switched on, on average
    (SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU), when the pass has the counter; and from the counting
    build of the kernel: obstacle cells within z_max per fetched lane-slot of the cell loop);
  * the algorithmic instruction floor: the fewest vector wave-instructions a particle needs on this hardware if
    every lane of every instruction did needed work --
        ALGO_PER_PARTICLE + ALGO_PER_CELL x cells_in_range / 64 + ALGO_PER_BEAM x beams / 64
    (the constants are counted out below, beside their definitions) -- priced with the kernel's
    average issue cost; `frac_of_algorithmic_floor` = that time / the measured kernel time.
"""
import hashlib
import json
import os
import re
import sys

# fewest vector instructions (per lane that does needed work) of the interval-only sector-query kernel
ALGO_PER_CELL = 41       # unpack + float32 coordinates and differences 4, exact float64 squared distance 6, range test 1,
                         # float32 bearing 20, place on the circle of intervals and distance to its edges 9, LDS address 1
ALGO_PER_BEAM = 30       # four float64 minima over the five intervals of a sector, root, the float32 mixture with two
                         # exponentials, the float64 product
ALGO_PER_PARTICLE = 50   # record (7 lane reads), clearing the minima, one logarithm of the lane's product, the exact lane sum, the result

SIMDS = 1024          # 256 CUs x 4 SIMD-32
CLOCK_GHZ = 2.4       # maximum shader clock

# opcodes per hardware sub-counter (SQ_INSTS_VALU_*)
BUCKETS = [
    ("ADD_F32", r"^v_(add|sub|subrev)_f32"), ("MUL_F32", r"^v_mul_f32"), ("FMA_F32", r"^v_(fma|fmac|fmaak|fmamk|mad|mac)_f32"),
    ("ADD_F64", r"^v_add_f64"), ("MUL_F64", r"^v_mul_f64"), ("FMA_F64", r"^v_(fma|fmac)_f64"),
    ("TRANS_F32", r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_f32"), ("TRANS_F64", r"^v_(rcp|rsq|sqrt)_f64"),
    ("CVT", r"^v_cvt_"),
    ("INT64", r"^v_(cmp\w*_[iu]64|lshlrev_b64|lshrrev_b64|ashrrev_i64|lshl_add_u64|mad_u64_u32|mad_i64_i32)"),
    ("INT32", r"^v_(add|sub|subrev|mul_lo|mul_hi|mul|mad|add3|lshl_add|lshl_or|and_or|or3|xad|min|max|cmp\w*|lshlrev|lshrrev|ashrrev|and|or|xor|not|bfe|bfi|bcnt|ffbh|ffbl|mbcnt\w*|sad|perm|alignbit|lshl|addc|subb|add_co|sub_co)_?(lo_|hi_)?(u32|i32|b32|u16|i16|b16|u32_b32|i32_i24|u32_u24|co_u32)"),
]


def base(mn):
    return re.sub(r"_(e32|e64|sdwa|dpp|e64_dpp)$", "", mn)


def cost_of(mn, table):
    """Issue cycles of one opcode: the measured table, else the rule the table shows (full rate only for the
    f32 add/mul/fma family, logic, add/sub and moves; 4 otherwise; 8 / 16 for transcendentals)."""
    b = base(mn)
    if mn.endswith("_dpp") or mn.endswith("_sdwa"):
        return 4.0
    if b == "v_cndmask_b32":
        return 2.0 if mn.endswith("_e32") else 4.0      # e32 priced from the measured compare + select pair
    if b in table:
        return float(round(table[b]))
    if re.match(r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_f32", b):
        return 8.0
    if re.match(r"^v_(rcp|rsq|sqrt)_f64", b):
        return 16.0
    if re.match(r"^v_div_(scale|fmas|fixup)_f64", b):
        return 4.0
    if re.match(r"^v_(add|sub|subrev|mul|fma|fmac|fmaak|fmamk)_f32$", b) or re.match(r"^v_(and|or|xor|add|sub|subrev)_(b32|u32)$", b) \
            or b in ("v_mov_b32", "v_ashrrev_i32", "v_not_b32"):
        return 2.0
    return 4.0


def bucket_of(mn):
    b = base(mn)
    for name, rx in BUCKETS:
        if re.match(rx, b):
            return name
    return "OTHER"


def static_mix(asm_path, kernel_sub, table):
    txt = open(asm_path).read()
    m = None
    for mm in re.finditer(r"^(_Z\w+):", txt, re.M):
        if kernel_sub in mm.group(1):
            m = mm
            break
    if m is None:
        raise SystemExit("kernel %r not found in %s" % (kernel_sub, asm_path))
    end = txt.index(".Lfunc_end", m.end())
    mix = {}
    for line in txt[m.end():end].splitlines():
        line = line.strip()
        mm = re.match(r"^(v_[a-z0-9_]+)", line)
        if not mm:
            continue
        mn = mm.group(1)
        if re.search(r"\b(row_shr|row_bcast|row_shl|quad_perm|row_mask)\b", line) and not mn.endswith("_dpp"):
            mn += "_dpp"
        if mn.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            pass
        e = mix.setdefault(mn, {"static_count": 0, "cycles": cost_of(mn, table), "bucket": bucket_of(mn)})
        e["static_count"] += 1
    return m.group(1), mix


def options(argv):
    """positional arguments, and the --name value ... options as a dict of lists"""
    pos, opt, cur = [], {}, None
    for a in argv:
        if a.startswith("--"):
            cur = a[2:]
            opt[cur] = []
        elif cur is None:
            pos.append(a)
        else:
            opt[cur].append(a)
    return pos, opt


def sha256_of(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def main():
    pos, opt = options(sys.argv[1:])
    pmc, cost, asm, ksub, out = pos[:5]
    pmc_name = pos[5] if len(pos) > 5 else "raycast_kernel<float, false, false, false"
    table = {k: v["cycles"] for k, v in json.load(open(cost))["cycles_per_wave64_instruction_per_simd"].items()}
    kname, mix = static_mix(asm, ksub, table)
    ctr = None
    for name, c in json.load(open(pmc)).items():
        if pmc_name in name:
            ctr = {k: v["per_launch_mean"] for k, v in c.items()}
            pmc_kernel = name
            break
    if ctr is None:
        raise SystemExit("no counters for the kernel in %s" % pmc)
    total = ctr["SQ_INSTS_VALU"]
    dyn = {b: ctr.get("SQ_INSTS_VALU_" + b, 0.0) for b, _ in BUCKETS}
    dyn["OTHER"] = total - sum(dyn.values())
    # average static cost per bucket
    avg, lo, hi = {}, {}, {}
    for b in list(dyn):
        ops = [e for e in mix.values() if e["bucket"] == b]
        n = sum(e["static_count"] for e in ops)
        avg[b] = sum(e["static_count"] * e["cycles"] for e in ops) / n if n else 4.0
        lo[b] = min((e["cycles"] for e in ops), default=4.0)
        hi[b] = max((e["cycles"] for e in ops), default=4.0)
    priced = sum(dyn[b] * avg[b] for b in dyn)
    priced_lo = sum(dyn[b] * lo[b] for b in dyn)
    priced_hi = sum(dyn[b] * hi[b] for b in dyn)
    res = {
        "kernel": pmc_kernel, "kernel_symbol": kname,
        "valu_wave_instructions_per_launch": total,
        "dynamic_count_by_bucket": dyn,
        "static_average_issue_cycles_by_bucket": avg,
        "priced_issue_cycles_per_launch": priced,
        "priced_issue_cycles_bounds": [priced_lo, priced_hi],
        "hardware_valu_active_cycles_per_launch": 4.0 * ctr.get("SQ_ACTIVE_INST_VALU", 0.0),
        "average_issue_cycles_per_instruction": priced / total,
        "simds": SIMDS, "peak_clock_GHz": CLOCK_GHZ,
        "floor_ms_at_peak_clock": priced / SIMDS / (CLOCK_GHZ * 1e9) * 1e3,
        "hbm_bytes_per_launch": (2.0 * ctr["FETCH_SIZE"] + ctr["WRITE_SIZE"]) * 1024.0 if "FETCH_SIZE" in ctr and "WRITE_SIZE" in ctr else None,
        "l2_hit_rate": ctr["TCC_HIT_sum"] / (ctr["TCC_HIT_sum"] + ctr["TCC_MISS_sum"]) if "TCC_HIT_sum" in ctr else None,
        "salu_instructions_per_launch": ctr.get("SQ_INSTS_SALU"), "lds_instructions_per_launch": ctr.get("SQ_INSTS_LDS"),
        "gpu_cycles_per_launch_GRBM_GUI_ACTIVE_div_8": ctr.get("GRBM_GUI_ACTIVE", 0.0) / 8.0,
        "static_opcode_mix": dict(sorted(mix.items(), key=lambda kv: -kv[1]["static_count"])),
        "note": "dynamic counts: rocprofv3 SQ_INSTS_VALU* per launch; cycles per opcode: tools/valu_issue_bench.hip on the same GPU; "
                "INT32 and OTHER (moves, selects, float compares, lane operations) are priced with the static opcode mix of the "
                "kernel's code; SQ_ACTIVE_INST_VALU counts quad-cycles",
    }
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if "sources" in opt:
        res["source_sha256"] = {os.path.relpath(os.path.abspath(f), root): sha256_of(f) for f in opt["sources"]}
    if "workload" in opt:
        res["workload"] = json.loads(" ".join(opt["workload"]))
    # lane utilisation by the hardware's count, when the passes have it
    if ctr.get("SQ_THREAD_CYCLES_VALU") and ctr.get("SQ_ACTIVE_INST_VALU"):
        res["lane_utilisation_hw"] = ctr["SQ_THREAD_CYCLES_VALU"] / (64.0 * ctr["SQ_ACTIVE_INST_VALU"])
        res["lane_utilisation_hw_is"] = "SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU), both counted per wave in quad-cycles: lanes switched on per vector instruction (lanes that compute for a cell outside the disc are on: what they do is not needed, see the cell loop's figure)"
    if "counts" in opt:
        res["cell_loop"] = json.load(open(opt["counts"][0]))
    if "work" in opt:
        particles, cells, beams = [float(x) for x in opt["work"][0].split(",")]
        per_particle = ALGO_PER_PARTICLE + ALGO_PER_CELL * cells / 64.0 + ALGO_PER_BEAM * beams / 64.0
        cpi = priced / total
        res["algorithmic_floor"] = {
            "particles": particles, "cells_in_range": cells, "beams": beams,
            "lane_instructions": {"per_cell_in_range": ALGO_PER_CELL, "per_beam": ALGO_PER_BEAM, "per_particle": ALGO_PER_PARTICLE},
            "wave_instructions_per_particle": per_particle,
            "wave_instructions_per_particle_issued": total / particles,
            "issue_cycles_per_instruction": cpi,
            "floor_ms_at_peak_clock": per_particle * particles * cpi / SIMDS / (CLOCK_GHZ * 1e9) * 1e3,
            "is": "every lane of every vector instruction doing needed work: no cell outside z_max fetched, no half-empty batch"}
    json.dump(res, open(out, "w"), indent=1)
    print("VALU wave-instructions per launch  %.4g" % total)
    for b in dyn:
        print("   %-10s %12.4g  x %.2f cycles" % (b, dyn[b], avg[b]))
    print("priced issue cycles per launch     %.4g   (bounds %.4g .. %.4g)" % (priced, priced_lo, priced_hi))
    print("hardware VALU-active cycles        %.4g" % res["hardware_valu_active_cycles_per_launch"])
    print("floor at %d SIMDs x %.1f GHz        %.3f ms" % (SIMDS, CLOCK_GHZ, res["floor_ms_at_peak_clock"]))


if __name__ == "__main__":
    main()
