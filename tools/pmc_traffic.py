#!/usr/bin/env python3
"""Post-process the rocprofv3 passes of tools/profile_bench.sh into small committed summaries:
  <out>/kernel_stats.csv       per-kernel calls / total / average duration (from the --stats pass)
  <out>/pmc_traffic.json       per-kernel HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (gfx950 correction of
                               MI355X_MICROARCH.md: FETCH_SIZE tallies 128-B requests at 64 B; counters are in KiB)"""
import collections
import csv
import glob
import json
import os
import sys

out, tag = sys.argv[1], sys.argv[2]
args = sys.argv[3:]

NAMES = {"se3d_k_normalize": "normalize", "se3d_k_pairs_flat": "epipolar_pairs", "se3d_k_pairs_packed": "epipolar_pairs",
         "se3d_k_postprocess_wave": "postprocess", "se3d_k_postprocess_wave2": "postprocess", "se3d_k_sweep_wave_lds": "sweep", "se3d_k_pairs_tiled": "epipolar_pairs",
         "se3d_k_sweep_fused_wave": "sweep", "se3d_k_sweep_fused": "sweep", "se3d_k_sweep_cost": "sweep_cost", "se3d_k_sweep_assign": "sweep_assign",
         "se3d_k_triangulate": "triangulate", "se3d_k_triangulate_hyp": "triangulate", "se3d_k_triangulate_slow": "triangulate_outliers", "se3d_k_postprocess": "postprocess", "se3d_k_stage2_walk": "stage2_walk",
         "se3d_k_track_cost": "track_cost", "se3d_k_track_assoc": "track_assoc", "se3d_k_lm_refine": "lm_refine",
         "se3d_k_track_near": "track_near", "se3d_k_track_update": "track_update"}


def short(name):
    n = name.replace("void ", "").split("(")[0].split("<")[0]
    return NAMES.get(n, n)


def newest(files):  # gpurun merges the outputs of successive runs into the same local directory: keep the latest run only
    return sorted(files, key=os.path.getmtime)[-1:] if files else []


def counters(sub, cname):
    files = newest(glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True))
    tot, cnt = collections.Counter(), collections.Counter()
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == cname:
                k = short(r["Kernel_Name"])
                tot[k] += float(r["Counter_Value"])
                cnt[k] += 1
    return tot, cnt


stats = newest(glob.glob(os.path.join(out, "stats", "**", "*kernel_stats.csv"), recursive=True))
if stats:
    rows = list(csv.DictReader(open(stats[0])))
    with open(os.path.join(out, "kernel_stats.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_ms", "avg_us", "percent"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], round(float(r["TotalDurationNs"]) / 1e6, 3), round(float(r["AverageNs"]) / 1e3, 2), r["Percentage"]])
# timed region only: per stream (context) a kernel is launched once in the set-up pass, W times in warm-up, K times in the timed region
# and once in the accounting pass (bench.py) -> launches [1+W, 1+W+K) of each stream, in start order
def arg(name, default):
    return int(args[args.index(name) + 1]) if name in args else default


K, W = arg("--steps", 20), arg("--warmup", 3)
trace = newest(glob.glob(os.path.join(out, "stats", "**", "*kernel_trace.csv"), recursive=True))
if trace:
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(trace[0])):
        per[(short(r["Kernel_Name"]), r["Stream_Id"])].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    agg = collections.defaultdict(list)
    for (k, st), v in per.items():
        if k.startswith("__amd"):
            continue
        v.sort()
        per_step = max(1, round(len(v) / (K + W + 2)))   # launches of this kernel per step on this stream
        sel = v[(1 + W) * per_step:(1 + W + K) * per_step] if k != "stage2_walk" else v[W * per_step:(W + K) * per_step]
        agg[k] += [(e - s_) / 1e3 for s_, e in sel]
    with open(os.path.join(out, "kernel_stats_timed_region.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "launches_in_timed_region", "avg_us", "min_us", "max_us"])
        for k, d in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
            if d:
                w.writerow([k, len(d), round(sum(d) / len(d), 2), round(min(d), 2), round(max(d), 2)])

ft, fc = counters("fetch", "FETCH_SIZE")
wt, wc = counters("write", "WRITE_SIZE")
res = {"workload": "%s: bench.py %s" % (tag, " ".join(args)),
       "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tools/profile_bench.sh",
       "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE tallies 128-B requests at 64 B; MI355X_MICROARCH.md, HBM)",
       "kernels": {}}
for k in sorted(set(ft) | set(wt)):
    if k.startswith("__amd"):
        continue
    f = ft[k] / max(fc[k], 1)
    w_ = wt[k] / max(wc[k], 1)
    res["kernels"][k] = {"fetch_size_kb_per_launch": round(f, 1), "write_size_kb_per_launch": round(w_, 1), "launches_sampled": int(fc[k]),
                         "hbm_bytes_per_launch_corrected": int((2 * f + w_) * 1024)}
json.dump(res, open(os.path.join(out, "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(res["kernels"].get("triangulate", {})))
