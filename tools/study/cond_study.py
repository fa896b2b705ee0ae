"""CPU study: oracle float32 JacobiSVD DLT vs exact fp64 null vector of the same float32 rows, per joint; which input-derived
quantity separates > 1 mm / > 1e-3 disagreements?"""
import sys, os, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from smartedgesensor3dhumanpose_amd import rig as R, scene as S
from oracle import oracle_py as O

def rows_f32(Rt, pts, weight):
    f = np.float32
    P = Rt.reshape(-1, 3, 4).astype(f)
    A = np.zeros((2 * len(P), 4), f)
    for i in range(len(P)):
        r0 = (pts[i, 0] * P[i, 2] - P[i, 0]).astype(f)
        r1 = (pts[i, 1] * P[i, 2] - P[i, 1]).astype(f)
        r0 = (r0 / np.sqrt((r0 * r0).sum(dtype=f), dtype=f)).astype(f)
        r1 = (r1 / np.sqrt((r1 * r1).sum(dtype=f), dtype=f)).astype(f)
        if weight:
            r0 = r0 * pts[i, 2]; r1 = r1 * pts[i, 2]
        A[2 * i] = r0; A[2 * i + 1] = r1
    return A

def solve64(A):
    u, s, vt = np.linalg.svd(A.astype(np.float64))
    v = vt[3]
    return v[:3] / v[3], s

def ut64(Rt, pts, cov2, mean):
    n = len(pts); dim = 2 * n; kappa = 0.5
    w = np.ones(2 * dim + 1); w[0] = 2 * kappa; w /= 2.0 * (dim + kappa)
    b = np.sqrt(np.float32(dim + kappa)).astype(np.float32)
    sp = [pts.copy() for _ in range(2 * dim + 1)]
    f = np.float32
    for c in range(n):
        a00, a10, a11 = f(cov2[c, 0]), f(cov2[c, 2]), f(cov2[c, 3])
        l11 = np.sqrt(a00); l21 = f(a10 / l11); l22 = np.sqrt(f(a11 - f(l21 * l21)))
        dx1 = f(l11 * b); dy1 = f(l21 * b); dy2 = f(l22 * b)
        sp[4 * c + 1][c, 0] -= dx1; sp[4 * c + 1][c, 1] -= dy1
        sp[4 * c + 2][c, 1] -= dy2
        sp[4 * c + 3][c, 0] += dx1; sp[4 * c + 3][c, 1] += dy1
        sp[4 * c + 4][c, 1] += dy2
    Xs = np.array([solve64(rows_f32(Rt, p, False))[0] for p in sp])
    d = Xs - mean.astype(np.float64)
    return (d * w[:, None]).T @ d

def main(seed, trials):
    rng = np.random.default_rng(seed)
    out = []
    for trial in range(trials):
        ncam = int(rng.choice([16, 16, 24]))
        rig = R.extr_rig_16cam() if ncam == 16 else R.elliptical_rig(24, seed=int(rng.integers(100)))
        rig.set_tensors(*O.rig_tensors(rig))
        persons = int(rng.integers(2, 8))
        extra = dict(p_miss=float(rng.uniform(*PMISS)), low_score_frac=float(rng.uniform(0, 0.3)), outlier_frac=float(rng.uniform(*OUTL)),
                     pixel_noise=float(rng.uniform(0.5, 4.0)), spacing=float(rng.uniform(0.5, 1.2)))
        frames, gt = S.generate_sequence(rig, S.SceneConfig(n_persons=persons, n_frames=2, seed=int(rng.integers(1 << 30)), **extra))
        K = rig.K.astype(np.float32)
        for fi, fr in enumerate(frames):
            ids = gt["ids"][fi]
            for p in range(persons):
                views = []
                for c in range(rig.n_cam):
                    w = np.nonzero(ids[c] == p)[0]
                    if len(w):
                        views.append((c, fr.persons[fr.det_offset[c] + w[0]]["keypoints"]))
                for j in range(17):
                    Rt, pts, cov2 = [], [], []
                    for c, kp in views:
                        if kp["score"][j] >= np.float32(0.3) and (kp["score"] >= np.float32(0.3)).sum() > 8:
                            fx, fy, cx, cy = K[c]
                            pts.append([(kp["x"][j] - cx) / fx, (kp["y"][j] - cy) / fy, kp["score"][j]])
                            cv = kp["cov"][j]
                            cov2.append([cv[0] / (fx * fx), cv[1] / (fx * fy), cv[1] / (fx * fy), cv[2] / (fy * fy)])
                            Rt.append(rig.Rt[c])
                    n = len(pts)
                    if n < 2:
                        continue
                    Rt = np.array(Rt, np.float32); pts = np.array(pts, np.float32); cov2 = np.array(cov2, np.float32)
                    Xo, _ = O.dlt(Rt, pts, True)
                    X64, s = solve64(rows_f32(Rt, pts, True))
                    _, su = solve64(rows_f32(Rt, pts, False))
                    rec = dict(n=n, err=float(np.linalg.norm(Xo - X64)), s=s.tolist(), su=su.tolist(), r=float(np.linalg.norm(X64)))
                    if DO_COV:
                        Co = O.ut_cov(Rt, pts, cov2, Xo).astype(np.float64)
                        C64 = ut64(Rt, pts, cov2, Xo)
                        rec["cov_rel"] = float(np.linalg.norm(Co - C64) / np.linalg.norm(C64))
                        rec["sig"] = float(np.sqrt(np.trace(C64)))
                    out.append(rec)
    return out

PMISS = tuple(float(x) for x in os.environ.get("PMISS", "0,0.15").split(","))
OUTL = tuple(float(x) for x in os.environ.get("OUTL", "0,0").split(","))
DO_COV = bool(int(os.environ.get("DO_COV", "0")))
if __name__ == "__main__":
    recs = main(int(sys.argv[1]), int(sys.argv[2]))
    json.dump(recs, open(sys.argv[3], "w"))
    e = np.array([r["err"] for r in recs])
    print(len(recs), "joints; err max mm", e.max() * 1e3, "over 1mm:", (e > 1e-3).sum(), "over 0.1mm", (e > 1e-4).sum())
