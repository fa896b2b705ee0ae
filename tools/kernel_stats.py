#!/usr/bin/env python3
"""rocprofv3 --kernel-trace --stats output -> small committed CSV: python3 tools/kernel_stats.py <rocprof dir> <out.csv>
Columns: kernel (raw name, template arguments kept), calls, total_ms, avg_us, min_us, max_us, percent."""
import csv
import glob
import os
import sys

out_dir, out_csv = sys.argv[1], sys.argv[2]
files = sorted(glob.glob(os.path.join(out_dir, "**", "*kernel_stats.csv"), recursive=True), key=os.path.getmtime)[-1:]
rows = list(csv.DictReader(open(files[0]))) if files else []
with open(out_csv, "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "percent"])
    for r in rows:
        name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        w.writerow([name, r["Calls"], round(float(r["TotalDurationNs"]) / 1e6, 3), round(float(r["AverageNs"]) / 1e3, 2),
                    round(float(r["MinNs"]) / 1e3, 2), round(float(r["MaxNs"]) / 1e3, 2), r["Percentage"]])
print(open(out_csv).read())
