#!/usr/bin/env python3
"""HBM bytes per launch from two separate rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; MI355X_MICROARCH.md: gfx950 FETCH_SIZE counts 128-B
requests at 64 B -> x2; counters are in KiB):  python3 tools/pmc_traffic2.py <fetch dir> <write dir> <out.json> <sequences> <command note>"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
fdir, wdir, out_json, seqs, note = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
import importlib.util
spec = importlib.util.spec_from_file_location("pmc_sq_names", os.path.join(os.path.dirname(os.path.abspath(__file__)), "pmc_sq.py"))


def short(name, _n={}):
    if not _n:
        src = open(spec.origin).read()
        _n.update(eval(src[src.index("NAMES = {") + 8: src.index("}", src.index("NAMES = {")) + 1]))
    n = name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0]
    return _n.get(n, n)


def counters(d, cname):
    files = sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True), key=os.path.getmtime)[-1:]
    tot, ids = collections.Counter(), collections.defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == cname:
                k = short(r["Kernel_Name"])
                tot[k] += float(r["Counter_Value"])
                ids[k].add(r["Dispatch_Id"])
    return tot, {k: len(v) for k, v in ids.items()}


ft, fc = counters(fdir, "FETCH_SIZE")
wt, wc = counters(wdir, "WRITE_SIZE")
res = {"workload": note, "sequences": seqs, "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), tools/profile_r3.sh",
       "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE tallies 128-B requests at 64 B for wide streaming reads; gather "
                     "kernels may be overstated by the x2; MI355X_MICROARCH.md, HBM)", "kernels": {}}
total = 0
# launches per pass of a kernel = its sampled launches / those of the post-processing kernel, which runs exactly once per stage-1 pass (round 6:
# round 5's total summed ONE launch per kernel although the pair kernel runs once per column chunk and the sweep kernels once per camera)
passes = max([fc.get(k, 0) for k in ("postprocess", "postprocess_block")] + [wc.get(k, 0) for k in ("postprocess", "postprocess_block")] + [1])
for k in sorted(set(ft) | set(wt)):
    if k.startswith("__amd"):
        continue
    f = ft[k] / max(fc.get(k, 1), 1)
    w_ = wt[k] / max(wc.get(k, 1), 1)
    b = int((2 * f + w_) * 1024)
    per_pass = max(fc.get(k, 0), wc.get(k, 0)) / passes
    if "reproj" not in k:   # the feedback reprojection follows the path; `pipeline` = one step of stage 1 + stage 2
        total += int(b * per_pass)
    res["kernels"][k] = {"fetch_size_kb_per_launch": round(f, 1), "write_size_kb_per_launch": round(w_, 1), "launches_sampled": int(fc.get(k, 0)),
                         "launches_per_pass": round(per_pass, 3), "hbm_bytes_per_launch_corrected": b}
res["passes_sampled"] = passes
res["pipeline_bytes_per_step_corrected"] = total   # sum over the kernels of bytes per launch x launches per pass
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smartedgesensor3dhumanpose_amd._lib import source_hash  # noqa: E402
res["source_hash"] = source_hash()   # the kernel sources these counts were collected on (bench.py uses the summary only for the same sources)
json.dump(res, open(out_json, "w"), indent=1)
print(json.dumps({k: v["hbm_bytes_per_launch_corrected"] for k, v in res["kernels"].items()}), total)
