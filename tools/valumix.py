#!/usr/bin/env python3
"""Vector-ALU instruction mix per kernel from the two PMC passes of `tools/profile_r3.sh valumix` and an issue-cycle model:
   python3 tools/valumix.py profiles/r3/pmc_valumix64_16cam6p_serial.json profiles/r3/pmc_valumix32_16cam6p_serial.json
Issue cost of a wave64 instruction on a SIMD (MI355X_MICROARCH.md: fp32 / int32 2 cycles; fp64 vector peak 78.6 TFLOP/s = 16 lanes per clock
= 4 cycles; transcendentals a quarter of that rate; conversions and 64-bit integer ops taken at 4; everything else — moves, compares,
selects, lane ops — at 2).  `issue_frac` = modelled issue cycles per SIMD / kernel cycles (GRBM_GUI_ACTIVE / 8 XCDs)."""
import json
import sys

a = json.load(open(sys.argv[1]))["kernels"]
b = json.load(open(sys.argv[2]))["kernels"]
COST = {"f64": 4, "trans_f64": 16, "f32": 2, "trans_f32": 8, "cvt": 4, "int32": 2, "int64": 4, "other": 2}
print("| kernel | ms | VALU instr (M) | fp64 | fp64 trans | cvt | fp32 | fp32 trans | int32 | int64 | other | issue_frac | fp64 share of kernel cycles | fp64 TFLOP/s |")
print("|---|---|---|---|---|---|---|---|---|---|---|---|---|---|")
for k in a:
    c = dict(a[k]["counters_per_launch"])
    c.update(b.get(k, {}).get("counters_per_launch", {}))
    g = lambda n: c.get("SQ_INSTS_VALU_" + n, 0)
    m = {"f64": g("ADD_F64") + g("MUL_F64") + g("FMA_F64"), "trans_f64": g("TRANS_F64"), "f32": g("ADD_F32") + g("MUL_F32") + g("FMA_F32"),
         "trans_f32": g("TRANS_F32"), "cvt": g("CVT"), "int32": g("INT32"), "int64": g("INT64")}
    tot = c.get("SQ_INSTS_VALU", 0)
    m["other"] = tot - sum(m.values())
    cyc = a[k]["kernel_cycles"]
    issue = sum(m[x] * COST[x] for x in m) / 1024.0
    flops64 = (g("ADD_F64") + g("MUL_F64") + 2 * g("FMA_F64")) * 64
    print("| %s | %.2f | %.0f | %s | %.3f | %.2f | %.1f |" % (k, a[k]["avg_us"] / 1e3, tot / 1e6, " | ".join("%.0f" % (m[x] / 1e6) for x in ("f64", "trans_f64", "cvt", "f32", "trans_f32", "int32", "int64", "other")),
                                                       issue / cyc, m["f64"] * 4 / 1024.0 / cyc, flops64 / (a[k]["avg_us"] * 1e-6) / 1e12))
