#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --pmc SQ_* pass (counter_collection.csv + kernel_trace.csv) per kernel:
  python3 tools/pmc_sq.py <rocprof output dir> <out.json> [note]
Per kernel (averages per launch): the raw counters, duration, and
  valu_issue_utilisation = SQ_INSTS_VALU x 2 cycles / (1024 SIMDs x kernel cycles)   (a wave64 fp32 / int32 VALU op holds a SIMD's issue for 2 cycles, fp64 for 4:
                                                                                     a LOWER bound for kernels with fp64 arithmetic, tools/valumix.py prices the mix;
                                                                                     kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs when collected,
                                                                                     else duration x 2.4 GHz)
  valu_busy   = 4 x SQ_ACTIVE_INST_VALU / (1024 x kernel cycles)    (SQ_* time counters tick in quad-cycles, MI355X_MICROARCH.md)
  wait_frac   = SQ_WAIT_ANY / SQ_WAVE_CYCLES, issue_stall_frac = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES, active_frac = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
out_dir, out_json = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""

NAMES = {"se3d_k_normalize": "normalize", "se3d_k_normalize_pairs": "epipolar_pairs", "se3d_k_front_fused": "front_fused", "se3d_k_pairs_rows": "epipolar_pairs", "se3d_k_postprocess_wave3": "postprocess", "se3d_k_pairs_flat": "epipolar_pairs", "se3d_k_pairs_packed": "epipolar_pairs",
         "se3d_k_postprocess_wave": "postprocess", "se3d_k_postprocess_wave2": "postprocess", "se3d_k_sweep_wave_lds": "sweep", "se3d_k_pairs_tiled": "epipolar_pairs_tiled", "se3d_k_sweep_fused_wave": "sweep", "se3d_k_sweep_fused": "sweep",
         "se3d_k_sweep_cost": "sweep_cost", "se3d_k_sweep_assign": "sweep_assign", "se3d_k_triangulate": "triangulate_joint", "se3d_k_triangulate_hyp": "triangulate",
         "se3d_k_triangulate_slow": "triangulate_outliers", "se3d_k_triangulate_exact": "triangulate_second_pass", "se3d_k_triangulate_exact4": "triangulate_second_pass_quad", "se3d_k_postprocess": "postprocess_block", "se3d_k_stage2_walk": "stage2_walk",
         "se3d_k_track_cost": "track_cost", "se3d_k_track_assoc": "track_assoc", "se3d_k_lm_refine": "lm_refine", "se3d_k_track_near": "track_near",
         "se3d_k_track_update": "track_update", "se3d_k_assoc_given": "assoc_given"}


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]
    return NAMES.get(n, n)


def newest(files):
    return sorted(files, key=os.path.getmtime)[-1:] if files else []


cc = newest(glob.glob(os.path.join(out_dir, "**", "*counter_collection.csv"), recursive=True))
tr = newest(glob.glob(os.path.join(out_dir, "**", "*kernel_trace.csv"), recursive=True))
tot = collections.defaultdict(collections.Counter)
disp = collections.defaultdict(set)
for f in cc:
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
dur = collections.defaultdict(list)
for f in tr:
    for r in csv.DictReader(open(f)):
        dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
res = {"source": "rocprofv3 --kernel-trace --pmc (tools/pmc_sq.py)", "note": note, "kernels": {}}
for k in sorted(tot):
    if k.startswith("__amd"):
        continue
    n = max(len(disp[k]), 1)
    c = {name: v / n for name, v in tot[k].items()}
    us = sum(dur[k]) / len(dur[k]) if dur[k] else 0.0
    cycles = c["GRBM_GUI_ACTIVE"] / 8.0 if c.get("GRBM_GUI_ACTIVE") else us * 1e-6 * 2.4e9
    e = {"launches": n, "avg_us": round(us, 2), "kernel_cycles": round(cycles), "counters_per_launch": {a: round(b) for a, b in sorted(c.items())}}
    if cycles > 0 and "SQ_INSTS_VALU" in c:
        e["valu_issue_utilisation"] = round(c["SQ_INSTS_VALU"] * 2.0 / (1024.0 * cycles), 4)
    if cycles > 0 and "SQ_ACTIVE_INST_VALU" in c:
        e["valu_busy"] = round(4.0 * c["SQ_ACTIVE_INST_VALU"] / (1024.0 * cycles), 4)
    if c.get("SQ_WAVE_CYCLES"):
        for name, key in (("wait_frac", "SQ_WAIT_ANY"), ("issue_stall_frac", "SQ_WAIT_INST_ANY"), ("active_frac", "SQ_ACTIVE_INST_ANY")):
            if key in c:
                e[name] = round(c[key] / c["SQ_WAVE_CYCLES"], 4)
    if us > 0 and c.get("GRBM_GUI_ACTIVE"):
        e["effective_clock_ghz"] = round(cycles / (us * 1e-6) / 1e9, 3)
    res["kernels"][k] = e
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smartedgesensor3dhumanpose_amd._lib import source_hash  # noqa: E402
res["source_hash"] = source_hash()   # the kernel sources these counts were collected on (bench.py uses the summary only for the same sources)
json.dump(res, open(out_json, "w"), indent=1)
print(json.dumps({k: {a: b for a, b in v.items() if a != "counters_per_launch"} for k, v in res["kernels"].items()}, indent=1))
