import json, sys, glob
import numpy as np
recs = []
for f in glob.glob(sys.argv[1]): recs += json.load(open(f))
recs = [r for r in recs if "cov_rel" in r and np.isfinite(r["cov_rel"])]
err = np.array([r["err"] for r in recs]); n = np.array([r["n"] for r in recs]); s = np.array([r["su"] for r in recs]); rr = np.array([r["r"] for r in recs])
cr = np.array([r["cov_rel"] for r in recs]); sig = np.array([r["sig"] for r in recs])
print(len(recs), "joints; cov_rel > 1e-3:", (cr > 1e-3).sum(), "max", cr.max(), "quantiles", np.quantile(cr, [0.5, 0.9, 0.99, 0.999]))
kap = (1 + rr**2) * s[:, 0] / (s[:, 2] - s[:, 3])
cands = {"eps*kap/sig": 6e-8 * kap / sig, "kap": kap, "1/sig": 1 / sig, "eps*(1+r^2)/sig": 6e-8*(1+rr**2)/sig, "eps*(1+r)/sig": 6e-8*(1+rr)/sig}
bad = cr > 1e-3
for k, v in cands.items():
    ratio = cr / v
    thr = v[bad].min() if bad.any() else np.inf
    print("%-18s min over bad %.3g flagged %d (%.3f%%); cr/v: median %.3g p99 %.3g max %.3g" % (k, thr, (v >= thr).sum(), 100 * (v >= thr).mean(), np.median(ratio), np.quantile(ratio, .99), ratio.max()))
i = np.argsort(-cr)[:10]
for j in i: print(n[j], "cr %.2e sig %.3g r %.3g kap %.3g err %.2e" % (cr[j], sig[j], rr[j], kap[j], err[j]), s[j])
