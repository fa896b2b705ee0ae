"""Shared helpers of the parity tests: compare a HIP stage-1/2 result with the oracle's on the same inputs."""
import numpy as np

from smartedgesensor3dhumanpose_amd import person_msgs as pm

POS_TOL_M = 1e-3        # north_star: per-joint error < 1 mm vs the CPU reference path (max, not mean)
SCORE_RTOL = 2e-5       # float32 mean-of-confidences; GPU sums in a different order
COV_REL_FRO = 1e-3      # covariances relative Frobenius <= 1e-3 (measured maxima are printed by the tests)


def normalized_detections(rig, frame):
    """numpy float32 restatement of row N (normalize_keypoints, S3D:312-333) for test-side bookkeeping."""
    out = []
    for c in range(rig.n_cam):
        fx, fy, cx, cy = [np.float32(v) for v in rig.K[c]]
        dets = []
        for d in range(frame.det_offset[c], frame.det_offset[c + 1]):
            kp = frame.persons[d]["keypoints"]
            valid = kp["score"] >= np.float32(0.30)
            if valid.sum() <= pm.NUM_KEYPOINTS // 2:
                continue
            a = np.zeros((pm.NUM_KEYPOINTS, 3), np.float32)
            a[:, 2] = -1.0
            a[valid, 0] = (kp["x"][valid] - cx) / fx
            a[valid, 1] = (kp["y"][valid] - cy) / fy
            a[valid, 2] = kp["score"][valid]
            dets.append((d - frame.det_offset[c], a))
        out.append(dets)
    return out


def compare_stage1(gpu, ora, check_cov=True):
    """Returns dict of max errors; asserts index-exactness."""
    assert gpu.flags == ora.flags, (gpu.flags, ora.flags)
    assert gpu.n_hyp == ora.n_hyp, (gpu.n_hyp, ora.n_hyp)
    assert np.array_equal(gpu.assoc(), ora.assoc()), "association table differs"
    assert gpu.n_persons == ora.n_persons, (gpu.n_persons, ora.n_persons)
    n = gpu.n_persons
    assert np.array_equal(gpu.person_hyp[:n], ora.person_hyp[:n])
    g, o = gpu.published()["keypoints"], ora.published()["keypoints"]
    stats = {"pos_max_mm": 0.0, "pos_mean_mm": 0.0, "cov_rel_max": 0.0, "n_joints": 0}
    if n == 0:
        return stats
    valid_g, valid_o = g["score"] > 0, o["score"] > 0
    assert np.array_equal(valid_g, valid_o), "set of valid joints differs"
    d = np.linalg.norm(g["joint"] - o["joint"], axis=-1)[valid_o]
    stats["pos_max_mm"] = float(d.max() * 1e3)
    stats["pos_mean_mm"] = float(d.mean() * 1e3)
    stats["n_joints"] = int(valid_o.sum())
    assert d.max() < POS_TOL_M, "max joint error %.4f mm" % (d.max() * 1e3)
    assert np.allclose(g["score"][valid_o], o["score"][valid_o], rtol=SCORE_RTOL, atol=1e-7)
    if check_cov:
        cg, co = g["cov"][valid_o], o["cov"][valid_o]
        nan_g, nan_o = np.isnan(cg).any(-1), np.isnan(co).any(-1)   # defined policy for non-SPD 2-D covariances: all six entries NaN
        assert np.array_equal(nan_g, nan_o), "NaN-covariance joints differ"
        assert np.isnan(cg[nan_g]).all() and np.isnan(co[nan_o]).all()
        stats["n_nan_cov"] = int(nan_o.sum())
        cg, co = cg[~nan_o], co[~nan_o]
        w = np.array([1, 2, 2, 1, 2, 1.0])
        rel = np.sqrt(((cg - co) ** 2 * w).sum(-1)) / np.sqrt((co ** 2 * w).sum(-1))
        stats["cov_rel_max"] = float(rel.max()) if len(rel) else 0.0
        assert stats["cov_rel_max"] < COV_REL_FRO, "covariance relative Frobenius error %.2e" % rel.max()
    return stats


REFINE_TOL_M = 1e-3     # refined joints: < 1 mm vs the CPU restatement (same GTSAM-equivalent LM semantics)
MARG_COV_RTOL = 1e-4    # marginal covariances are fp64 on both sides (+ what the measured joint difference implies: compare_stage2)


def compare_stage2(gpu, ora, check_iters=True, check_tracks=True):
    """gpu/ora: FusedResult of one pose_prior callback.  check_tracks=False for results fetched after a fused multi-step walk: the
    track count is sequence state (what it is after the last step of the walk), not a per-step output."""
    assert not check_tracks or gpu.c.n_tracks == ora.c.n_tracks, (gpu.c.n_tracks, ora.c.n_tracks)
    assert gpu.n_persons == ora.n_persons, (gpu.n_persons, ora.n_persons)
    assert abs(gpu.c.pred_delta_t - ora.c.pred_delta_t) < 1e-7
    n = gpu.n_persons
    stats = {"pos_max_mm": 0.0, "pred_max_mm": 0.0, "cov_rel_max": 0.0}
    if n == 0:
        return stats
    assert np.array_equal(gpu.fused["id"][:n], ora.fused["id"][:n])
    assert np.array_equal(gpu.fused_pred["id"][:n], ora.fused_pred["id"][:n])
    for name, tolkey in (("fused", "pos_max_mm"), ("fused_pred", "pred_max_mm")):
        g, o = getattr(gpu, name)[:n]["keypoints"], getattr(ora, name)[:n]["keypoints"]
        vg, vo = g["score"] > 0, o["score"] > 0
        assert np.array_equal(vg, vo)
        assert np.array_equal(g["score"], o["score"])  # max(0.10, measured score): exact
        d = np.linalg.norm(g["joint"] - o["joint"], axis=-1)[vo]
        stats[tolkey] = float(d.max() * 1e3)
        assert d.max() < REFINE_TOL_M, "%s max joint error %.5f mm" % (name, d.max() * 1e3)
        cg, co = g["cov"][vo], o["cov"][vo]
        rel = np.abs(cg - co).max(-1) / np.abs(co).max(-1)
        stats["cov_rel_max"] = max(stats["cov_rel_max"], float(rel.max()))
        # A marginal covariance is a block of (J^T J)^-1 at the optimum; the range factors' Jacobians are unit bone directions, which turn by
        # delta / L when a joint moves by delta (L >= 0.05 m: the shortest bone, PP:386-479): two optima delta apart — inside the position
        # tolerance — differ by up to ~2 delta / L in the covariance.  The tolerance allows for the measured delta of this result.
        cov_tol = MARG_COV_RTOL + 2.0 * float(d.max()) / 0.05
        assert rel.max() < cov_tol, "%s covariance rel error %.2e (tolerance %.2e at a joint difference of %.2e m)" % (name, rel.max(), cov_tol, d.max())
    if check_iters:
        assert np.array_equal(gpu.lm_iterations[:n], ora.lm_iterations[:n]), (gpu.lm_iterations[:n], ora.lm_iterations[:n])
    return stats


# ---------------------------------------------------------------------------------------------------------------------------------
# Ill-conditioning criteria of the randomised sweep (tests/test_gpu_fuzz_parity.py), computed FROM THE INPUTS in float64.
#
# Rows T/U: the reference takes the DLT null vector from a float32 JacobiSVD (S3D:456), the kernels from an fp64 solve of the same
# float32 rows.  A float32 perturbation of relative size eps = 2^-24 of the rows turns the null vector by ~ eps * s1 / (s3 - s4)
# (singular values of the 2n x 4 system) and dehomogenising v -> v[:3] / v[3] multiplies that by ~ (1 + |X|^2): a joint whose system has
#     kappa = (1 + |X|^2) * s1 / (s3 - s4)
# can come out of the two solvers up to ~ c * eps * kappa metres apart (measured c <= 5 over 3e5 joints, tools: profiles/r3/README.md).
# kappa >= DLT_KAPPA_MAX is the exclusion set of the < 1 mm / 1e-3 assertions: such joints are triangulated far outside the hall from
# nearly parallel or inconsistent rays (gross 2-D outliers, wrong associations) and the reference's own answer there is rounding noise.
# ---------------------------------------------------------------------------------------------------------------------------------
BRANCH_TIE_REL = 1.0e-6   # relative distance of a reprojection error from a decision threshold of the outlier rejection below which the joint is excluded
DLT_KAPPA_MAX = 2.0e3     # c * eps * kappa = 5 * 6e-8 * 2e3 = 0.6 mm: below it the two solvers must agree to POS_TOL_M


def cov_tolerance(kappa, cov):
    """Relative-Frobenius tolerance of a stage-1 covariance: COV_REL_FRO + 2 * 2^-24 * kappa / sigma, sigma = sqrt(mean diagonal)
    of the oracle's covariance `cov` [..., 6].  The unscented transform differences 4n + 1 sample solutions that lie ~sigma apart; the oracle
    (like the reference) solves every sample in float32, whose noise on a sample is c 2^-24 kappa metres (c: median 0.31, maximum 4.91 over
    3.1e5 joints, profiles/r3/cond_study.txt), so its covariance carries a relative error of about 2 c 2^-24 kappa / sigma — the kernels' fp64
    solves have none of it (CPU study of 46 k joints, profiles/r3/cond_study.txt: the oracle's deviation from the exact unscented transform divided
    by 2^-24 kappa / sigma has median 0.27 and 99th percentile 1.45; its far tail is the large-sigma joints of covariance_in_hall).  The term is below 1e-4 for the bulk of the joints (kappa ~ 20, sigma ~ 2 cm) and decides the rare tail: over
    86 seeds x 40 scenes (9.5 M asserted joints) three joints exceed the flat 1e-3 for this reason — 1.46e-3 at kappa 62 / sigma 4 mm (19
    views), 1.01e-3 at kappa 744, 1.2e-3 — all inside this tolerance (profiles/r3/README.md)."""
    cov = np.asarray(cov, np.float64)
    with np.errstate(invalid="ignore"):
        sigma = np.sqrt(np.maximum((cov[..., 0] + cov[..., 3] + cov[..., 5]) / 3.0, 1e-300))
    return COV_REL_FRO + 2.0 * 2.0 ** -24 * np.asarray(kappa, np.float64) / sigma


COV_SIGMA_MAX_M = 1.0   # covariance assertion: joints whose oracle covariance stays below (1 m)^2 on the diagonal


def covariance_in_hall(cov):
    """bool [...]: the covariance `cov` [..., 6] says the joint is known to better than COV_SIGMA_MAX_M.  The unscented transform's samples
    of a joint with a larger covariance land metres to tens of metres away from where the rays converge; THOSE systems are ill-conditioned
    (kappa grows with 1 + |X|^2) whatever the base system's kappa is, and the reference's float32 solves of them are noise at the 1e-3 level
    (seen at 32 / 48 cameras with 12 - 24 persons: spurious persons with sigma 30 - 50 m, base kappa 59 - 1150, covariances 1.3e-3 - 7.9e-3 apart,
    positions within 0.05 mm).  Such joints leave the covariance assertion (counted), not the position / score assertions."""
    cov = np.asarray(cov, np.float64)
    with np.errstate(invalid="ignore"):
        return np.nan_to_num(np.maximum(np.maximum(cov[..., 0], cov[..., 3]), cov[..., 5]), nan=np.inf) < COV_SIGMA_MAX_M ** 2


def covariance_condition(kappa, tables):
    """kappa [n][17] of the systems a joint's COVARIANCE depends on: its own and — through the limb-length term added to the diagonal, a
    function of the distance to the parent joint (S3D:866-872; both shoulders in the "simple" model, S3D:875-881) — its parent's.  `kappa` as
    returned by published_joint_condition (merge partners included, also for joints the published person does not carry: a partner's parent
    joint may have been removed by the root-distance filter after it inflated the child's covariance; seen: a sane ear with a covariance
    of (94 km)^2 x I from a partner whose eye lay 62 km away)."""
    par, L = tables["kpParent"], tables["limbLength"]
    out = np.array(kappa, np.float64, copy=True)
    for j in range(pm.NUM_KEYPOINTS):
        if par[j] >= 0 and L[j] > 0:
            out[:, j] = np.fmax(out[:, j], kappa[:, par[j]])
    if tables.get("shoulderDist", 0) > 0 and par[6] >= 0 and not (L[6] > 0):
        both = np.fmax(kappa[:, 5], kappa[:, 6])
        out[:, 5], out[:, 6] = np.fmax(out[:, 5], both), np.fmax(out[:, 6], both)
    return out


GARBAGE_JOINT_DIST_M = 10.0   # stage 2: a measured joint further than this from the median joint of its person (no body spans 10 m)
GARBAGE_JOINT_SIGMA_M = 1.0e2  # stage 2: a measured joint whose covariance says "unknown to more than 100 m" (see garbage_persons)


def _dlt_rows(rig, frame, assoc_rows):
    """float32 DLT rows of every (person, joint, camera) exactly as S3D:446-453 builds them, + usable-view mask and confidences."""
    f = np.float32
    n, N = len(assoc_rows), rig.n_cam
    K = rig.K.astype(f)
    P = rig.Rt.reshape(N, 3, 4).astype(f)
    x = np.zeros((n, pm.NUM_KEYPOINTS, N), f)
    y = np.zeros_like(x)
    cf = np.full_like(x, -1.0)
    for c in range(N):
        idx = assoc_rows[:, c]
        m = idx >= 0
        if not m.any():
            continue
        kp = frame.persons[frame.det_offset[c] + idx[m]]["keypoints"]
        ok = kp["score"] >= f(0.30)                      # normalize_keypoints gate (S3D:321) == triangulation gate (S3D:725)
        x[m, :, c] = np.where(ok, (kp["x"] - K[c, 2]) / K[c, 0], f(0))
        y[m, :, c] = np.where(ok, (kp["y"] - K[c, 3]) / K[c, 1], f(0))
        cf[m, :, c] = np.where(ok, kp["score"], f(-1))
    use = cf >= f(0.30)
    r0 = (x[..., None] * P[None, None, :, 2, :] - P[None, None, :, 0, :]).astype(f)
    r1 = (y[..., None] * P[None, None, :, 2, :] - P[None, None, :, 1, :]).astype(f)
    r0 = (r0 / np.sqrt((r0 * r0).sum(-1, dtype=f), dtype=f)[..., None]).astype(f)
    r1 = (r1 / np.sqrt((r1 * r1).sum(-1, dtype=f), dtype=f)[..., None]).astype(f)
    return r0, r1, use, np.where(use, cf, f(0)), x, y


def _kappa_from_gram(G):
    """G [..., 4, 4] fp64 Gram matrix A^T A -> (kappa, X [..., 3])."""
    lam, vec = np.linalg.eigh(G)
    s = np.sqrt(np.maximum(lam, 0.0))
    v = vec[..., :, 0]
    w = np.where(np.abs(v[..., 3:4]) > 1e-150, v[..., 3:4], 1e-150)
    X = v[..., :3] / w
    gap = np.maximum(s[..., 1] - s[..., 0], 1e-150)
    with np.errstate(over="ignore", invalid="ignore"):
        kappa = (1.0 + (X * X).sum(-1)) * s[..., 3] / gap
    return np.where(np.isfinite(kappa), kappa, np.inf), X


def dlt_condition(rig, frame, assoc_rows, want_points=False):
    """kappa [n][17] of the DLT systems behind each (hypothesis, joint) of a frame (nan: fewer than two usable views), the maximum over the
    confidence-weighted system of the main solve (S3D:746), the unweighted system of the sigma points (S3D:518) and — where the fp64
    reprojection error comes near g_reproj_error_max_acceptable, so that the outlier rejection (S3D:748-838) may have removed a view —
    their leave-one-view-out subsets.  assoc_rows: [n][n_cam] original detection indices or -1 (rows of the result's association table).
    want_points: also the fp64 solution of the weighted system [n][17][3]."""
    assoc_rows = np.asarray(assoc_rows, np.int32).reshape(-1, rig.n_cam)
    r0, r1, use, cf, x, y = _dlt_rows(rig, frame, assoc_rows)
    a0, a1 = r0.astype(np.float64), r1.astype(np.float64)
    outer = a0[..., :, None] * a0[..., None, :] + a1[..., :, None] * a1[..., None, :]          # [n,17,N,4,4]
    u = use.astype(np.float64)
    w2 = (cf.astype(np.float64)) ** 2
    Gu = (outer * u[..., None, None]).sum(2)
    Gw = (outer * w2[..., None, None]).sum(2)
    nv = use.sum(-1)
    ku, _ = _kappa_from_gram(Gu)
    kw, X = _kappa_from_gram(Gw)
    kappa = np.maximum(ku, kw)
    # reprojection error of the weighted solution (S3D:425-438) in fp64
    P = rig.Rt.reshape(rig.n_cam, 3, 4).astype(np.float64)
    Xh = np.concatenate([X, np.ones(X.shape[:-1] + (1,))], -1)
    pr = np.einsum("cij,pkj->pkci", P, Xh)
    z = np.where(np.abs(pr[..., 2]) > 1e-150, pr[..., 2], 1e-150)
    with np.errstate(over="ignore", invalid="ignore"):
        e = np.hypot(pr[..., 0] / z - x, pr[..., 1] / z - y)
        cfd = cf.astype(np.float64)
        err = (cfd * np.where(use, e, 0.0)).sum(-1) / np.maximum(cfd.sum(-1), 1e-150)
    def sub_err(p, k, Xc, keep):     # confidence-weighted reprojection error of a point over a subset of the joint's views (S3D:425-438), fp64
        prc = P @ np.append(Xc, 1.0)
        zc = np.where(np.abs(prc[:, 2]) > 1e-150, prc[:, 2], 1e-150)
        with np.errstate(over="ignore", invalid="ignore"):
            ec = np.hypot(prc[:, 0] / zc - x[p, k], prc[:, 1] / zc - y[p, k])
        w = cfd[p, k] * keep
        return float((w * np.where(keep > 0, ec, 0.0)).sum() / max(w.sum(), 1e-150))
    for p, k in zip(*np.nonzero(~(err <= 0.8 * 0.05) & (nv >= 3))):
        errs_loo = []
        for c in np.nonzero(use[p, k])[0]:
            kk_u, _ = _kappa_from_gram(Gu[p, k] - outer[p, k, c])
            kk_w, Xc = _kappa_from_gram(Gw[p, k] - outer[p, k, c] * w2[p, k, c])
            kappa[p, k] = max(kappa[p, k], kk_u, kk_w)
            keep = use[p, k].astype(np.float64)
            keep[c] = 0.0
            errs_loo.append(sub_err(p, k, Xc, keep))
        # Branch ties (the reference's decisions are discontinuous): the error against g_reproj_error_max_acceptable (S3D:748), the best
        # leave-one-view-out error against 0.9 x error (S3D:830), two candidates for "best".  Within BRANCH_TIE_REL of a threshold the float32
        # terms of the two sides decide differently and the results are different points (seen: err = 0.050000002, 368 mm) — excluded.
        e0 = float(err[p, k])
        tie = abs(e0 - 0.05) <= BRANCH_TIE_REL * 0.05
        if nv[p, k] >= 4 and e0 > 0.05 * (1.0 - BRANCH_TIE_REL):
            se = np.sort(np.array(errs_loo))
            tie = tie or abs(se[0] - 0.9 * e0) <= BRANCH_TIE_REL * 0.9 * e0
            if len(se) > 1 and se[0] < 0.9 * e0 * (1.0 + BRANCH_TIE_REL):
                tie = tie or (se[1] - se[0]) <= BRANCH_TIE_REL * se[0]
        if tie:
            kappa[p, k] = np.inf
    kappa = np.where(nv >= 2, kappa, np.nan)
    return (kappa, X) if want_points else kappa


def published_joint_condition(rig, frame, result, k2f=None):
    """kappa [n_persons][17] for the PUBLISHED persons of a stage-1 result (keypoint order of the input, not the fusion layout).  A published
    person is its own hypothesis or, after the duplicate merge (S3D:984-996: the later of two persons closer than 0.20 m is averaged into
    the earlier one and erased, joints only one of them has are taken over), a blend: every unpublished later hypothesis whose fp64 joints
    lie within 0.5 m (median over the joints both have; the reference's test runs after its plausibility filters, which this criterion
    does not redo) or that has one joint within 2 cm (a joint taken over unchanged) counts as a possible merge partner and contributes the maximum of its kappa."""
    n = result.n_persons
    hyp = result.person_hyp[:n]
    kap, X = dlt_condition(rig, frame, result.assoc(), want_points=True)
    out = kap[hyp].copy()
    pub = np.zeros(len(kap), bool)
    pub[hyp] = True
    k2f = np.array(pm.KP2FUSION_SIMPLE if k2f is None else k2f)   # input keypoint -> fusion slot of the pose method
    kp = result.published()["keypoints"][:, k2f]
    for h2 in np.nonzero(~pub)[0]:
        have2 = ~np.isnan(kap[h2])
        for i in range(n):
            if hyp[i] > h2:
                break
            both = have2 & (kp["score"][i] > 0)
            if not both.any():
                continue
            dist = np.linalg.norm(X[h2][both] - kp["joint"][i][both], axis=-1)
            if np.median(dist) < 0.5 or dist.min() < 0.02:     # close as a whole, or a joint that was taken over as it is
                out[i] = np.where(have2, np.fmax(out[i], kap[h2]), out[i])
    # A published joint that neither the person's own hypothesis nor a recognised partner carries (crowded scenes: chains of merges, a
    # partner whose fp64 points lie far from the float32 ones) was taken over from SOME later unpublished hypothesis: the largest kappa among
    # those that have the joint (conservative: the joint is asserted only if all candidates are well-conditioned)
    for i in range(n):
        for j in np.nonzero(np.isnan(out[i]) & (kp["score"][i] > 0))[0]:
            cand = [kap[h2][j] for h2 in np.nonzero(~pub)[0] if h2 > hyp[i] and not np.isnan(kap[h2][j])]
            if cand:
                out[i, j] = max(cand)
    return out


def score_allowance(rig, frame, gpu, ora, k2f=None):
    """Relative allowance [n_persons][17] for the comparison of stage-1 SCORES that comes from the position difference of the two sides.
    A joint's score is the mean confidence of its views times min(1, g_reproj_error_max_acceptable / err) (S3D:840-844): once the
    reprojection error err exceeds 0.05 the score carries err's dependence on the triangulated point, and err = mean |proj(X) - x| changes by
    d / depth per metre of position difference (2e-4 of itself for 0.05 mm at 5 m and err = 0.05; without bound where a ray passes close to a
    camera centre: err = 74 seen on a two-view joint).  Returned: |err(X_gpu) - err(X_oracle)| / min(...) in fp64 over the usable views of the
    person's hypothesis — maximum over the full set and, with three or more views, its leave-one-view-out subsets (the outlier rejection may have
    removed one, S3D:748-838) — where any of those errors comes within 20 % of the threshold or above it; 0 elsewhere (score = mean confidence)."""
    n = gpu.n_persons
    k2f = np.array(pm.KP2FUSION_SIMPLE if k2f is None else k2f)
    rows = gpu.assoc()[gpu.person_hyp[:n]]
    _, _, use, cf, x, y = _dlt_rows(rig, frame, rows)
    P = rig.Rt.reshape(rig.n_cam, 3, 4).astype(np.float64)
    cfd = np.where(use, cf.astype(np.float64), 0.0)
    errs = []
    for side in (gpu, ora):
        X = side.published()["keypoints"]["joint"][:, k2f]
        Xh = np.concatenate([X, np.ones(X.shape[:-1] + (1,))], -1)
        pr = np.einsum("cij,pkj->pkci", P, Xh)
        z = np.where(np.abs(pr[..., 2]) > 1e-150, pr[..., 2], 1e-150)
        with np.errstate(over="ignore", invalid="ignore"):
            errs.append(cfd * np.where(use, np.hypot(pr[..., 0] / z - x, pr[..., 1] / z - y), 0.0))     # [n][17][N] confidence-weighted terms
    out = np.zeros((n, pm.NUM_KEYPOINTS))
    nv = use.sum(-1)
    with np.errstate(invalid="ignore", divide="ignore"):
        for drop in [None] + list(range(rig.n_cam)):
            keep = np.ones(rig.n_cam, bool)
            if drop is not None:
                keep[drop] = False
            den = np.maximum((cfd * keep).sum(-1), 1e-150)
            eg, eo = (errs[0] * keep).sum(-1) / den, (errs[1] * keep).sum(-1) / den
            lo = np.minimum(eg, eo)
            rel = np.where(lo > 0, np.abs(eg - eo) / lo, 0.0)
            applies = (np.maximum(eg, eo) > 0.8 * 0.05) & (nv >= (2 if drop is None else 3)) & (True if drop is None else use[..., drop])
            out = np.where(applies, np.fmax(out, rel), out)
    return np.nan_to_num(out, nan=0.0, posinf=np.inf)


def garbage_persons(persons3d):
    """bool [n]: stage-1 persons with a valid joint more than GARBAGE_JOINT_DIST_M from the median of their valid joints — a joint
    triangulated from an inconsistent pair of rays on a person without a root (no root-distance filter, S3D:937-953).  Its LM problem
    (PP:690-746) is flat along that ray and badly scaled; algebraically equivalent solvers take different accept / reject paths.
    The same holds for a valid joint that sits near the body but whose covariance says it is unknown to more than GARBAGE_JOINT_SIGMA_M (an
    ill-conditioned DLT system): the information of its unary factor (PP:690), 1 / sigma^2 < 1e-4 m^-2, is more than seven orders of magnitude
    below that of the joint's bone factor (~1e3 m^-2), so the joint's 3 x 3 block of the normal equations has a condition number above 1e7 and
    its inverse — which every elimination order forms — carries the rounding of the large entries into the two unconstrained directions.  Seen:
    sigma 90 km (condition 9e12) — two algebraically equal LM runs differ in THAT joint by 5.8 mm while every other joint of the person agrees to
    0.006 mm; sigma 761 m — positions equal to 1e-14 m, the joint's marginal covariance differs by 1.2e-4 (tolerance 1e-4)."""
    out = np.zeros(len(persons3d), bool)
    for i, p in enumerate(persons3d):
        v = p["keypoints"]["score"] > 0
        if v.sum() >= 2:
            J = p["keypoints"]["joint"][v]
            far = np.linalg.norm(J - np.median(J, 0), axis=-1).max() > GARBAGE_JOINT_DIST_M
            var = p["keypoints"]["cov"][v][:, [0, 3, 5]]
            with np.errstate(invalid="ignore"):
                unknown = bool((np.nan_to_num(var, nan=0.0, posinf=np.inf).max() > GARBAGE_JOINT_SIGMA_M ** 2))
            out[i] = far or unknown
    return out


def inflation_allowance(tables, joints, valid, d):
    """Absolute Frobenius allowance [n][17] for the comparison of stage-1 covariances that comes from the POSITION difference d [n][17] (m)
    between the two sides: the limb-length term added to the diagonal (S3D:861-883), t = (g_limbLCovOffsetSigma (dist - L) / sigma_L)^2, is
    a function of the joint positions — a difference of nearly equal numbers when the limb has its model length — so two results whose
    joints differ by d (inside the 1 mm tolerance) differ in t by |dt| <= 2 c^2 |dist - L| D + c^2 D^2 with D <= d_joint + d_parent,
    c = 0.075 / sigma_L, on each of the three diagonal entries.  joints / valid: one side's positions and validity in keypoint order;
    tables: oracle.pose_tables(pose_method)."""
    par, L, sg = tables["kpParent"], tables["limbLength"], tables["limbLSigma"]
    out = np.zeros(valid.shape)

    def term(j, k, length, sigma, targets):
        both = valid[:, j] & valid[:, k]
        dist = np.linalg.norm(joints[:, j] - joints[:, k], axis=-1)
        D = d[:, j] + d[:, k]
        c2 = (0.075 / sigma) ** 2
        a = np.sqrt(3.0) * (2.0 * c2 * np.abs(dist - length) * D + c2 * D * D)
        for t_ in targets:
            out[:, t_] += np.where(both, a, 0.0)
    for j in range(pm.NUM_KEYPOINTS):
        if par[j] >= 0 and L[j] > 0:
            term(j, par[j], L[j], sg[j], [j])
    if tables.get("shoulderDist", 0) > 0:   # "simple": the right shoulder's term goes to both shoulders (S3D:875-881); keypoints 6 / 5
        rs, ls = 6, 5
        if par[rs] >= 0 and not (L[rs] > 0):
            term(rs, ls, tables["shoulderDist"], tables["shoulderSigma"], [rs, ls])
    return out
