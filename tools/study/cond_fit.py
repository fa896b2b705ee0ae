import json, sys, glob
import numpy as np
recs = []
for f in glob.glob(sys.argv[1]): recs += json.load(open(f))
err = np.array([r["err"] for r in recs]); n = np.array([r["n"] for r in recs]); s = np.array([r["s"] for r in recs]); rr = np.array([r["r"] for r in recs])
print(len(recs), "joints;", (err > 1e-3).sum(), "over 1mm; by n:", {k: int(((err > 1e-3) & (n == k)).sum()) for k in range(2, 8)})
cands = {
 "s1/s3": s[:, 0] / s[:, 2],
 "s1/(s3-s4)": s[:, 0] / (s[:, 2] - s[:, 3]),
 "(1+r^2)*s1/s3": (1 + rr**2) * s[:, 0] / s[:, 2],
 "(1+r^2)*s1/(s3-s4)": (1 + rr**2) * s[:, 0] / (s[:, 2] - s[:, 3]),
 "(1+r)*s1/s3": (1 + rr) * s[:, 0] / s[:, 2],
 "r": rr,
}
bad = err > 1e-3
for k, v in cands.items():
    thr = v[bad].min()
    # predicted error model: err ~ c * eps * v ; find c range
    ratio = err / (v * 6e-8)
    print("%-22s min over bad %.3g -> flagged %d (%.3f%%) ; c=err/(eps*v): median %.3g p99 %.3g max %.3g" % (k, thr, (v >= thr).sum(), 100 * (v >= thr).mean(), np.median(ratio), np.quantile(ratio, 0.99), ratio.max()))
