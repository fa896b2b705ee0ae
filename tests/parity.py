"""Shared helpers of the parity tests: compare a HIP stage-1/2 result with the oracle's on the same inputs."""
import numpy as np

from smartedgesensor3dhumanpose_amd import person_msgs as pm

POS_TOL_M = 1e-3        # north_star: per-joint error < 1 mm vs the CPU reference path (max, not mean)
SCORE_RTOL = 0.0        # round 4 (late): the kernels form the score with the reference's float expression (S3D:730, 738) -> bit-equal
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
    assert np.array_equal(g["score"][valid_o], o["score"][valid_o]), "scores differ (max rel %.3g)" % float(np.abs(g["score"][valid_o] / o["score"][valid_o] - 1).max())
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
        # per person (a bone's two ends belong to one person), capped: 1e-2 is what a joint difference of 0.25 mm explains — a larger covariance
        # deviation is a finding even when the positions are inside their tolerance.
        dfull = np.where(vo, np.linalg.norm(g["joint"] - o["joint"], axis=-1), 0.0)
        cov_tol = MARG_COV_RTOL + np.minimum(2.0 * dfull.max(-1, keepdims=True) / 0.05, 1e-2) * np.ones_like(dfull)
        assert (rel < cov_tol[vo]).all(), "%s covariance rel error %.2e (tolerance %.2e at a joint difference of %.2e m)" % (name, rel.max(), cov_tol[vo][np.argmax(rel)], d.max())
    if check_iters:
        assert np.array_equal(gpu.lm_iterations[:n], ora.lm_iterations[:n]), (gpu.lm_iterations[:n], ora.lm_iterations[:n])
    return stats


def classify_stage2(gpu, ora):
    """What kind of difference compare_stage2 would trip over first (round 6: the mismatches inside stage 2's exempt set are counted by kind):
    "count" (published persons or live tracks), "ids", "valid" (sets of valid joints / scores), "fused" (a refined joint >= 1 mm apart),
    "pred" (refined joints inside 1 mm, a PREDICTED joint >= 1 mm apart: the two sides carry different velocity histories), "cov", "iters", or None."""
    if gpu.n_persons != ora.n_persons or gpu.c.n_tracks != ora.c.n_tracks:
        return "count"
    n = gpu.n_persons
    if n == 0:
        return None
    if not (np.array_equal(gpu.fused["id"][:n], ora.fused["id"][:n]) and np.array_equal(gpu.fused_pred["id"][:n], ora.fused_pred["id"][:n])):
        return "ids"
    cov = False
    for name in ("fused", "fused_pred"):
        g, o = getattr(gpu, name)[:n]["keypoints"], getattr(ora, name)[:n]["keypoints"]
        if not np.array_equal(g["score"], o["score"]):
            return "valid"
        vo = o["score"] > 0
        dfull = np.where(vo, np.linalg.norm(g["joint"] - o["joint"], axis=-1), 0.0)
        if dfull.max() >= REFINE_TOL_M:
            return "fused" if name == "fused" else "pred"
        rel = np.abs(g["cov"][vo] - o["cov"][vo]).max(-1) / np.abs(o["cov"][vo]).max(-1)
        tol = (MARG_COV_RTOL + np.minimum(2.0 * dfull.max(-1, keepdims=True) / 0.05, 1e-2) * np.ones_like(dfull))[vo]   # compare_stage2's tolerance
        cov = cov or bool((rel >= tol).any())
    if not np.array_equal(gpu.lm_iterations[:n], ora.lm_iterations[:n]):
        return "iters" if not cov else "cov"
    return "cov" if cov else None


OBJECTIVE_RTOL = 1e-9   # stage 2: where the two LMs publish different joints, the library's final objective may exceed the oracle's by this much at most


def compare_stage2_objectives(gpu, ora):
    """Solver-independent check of one pose_prior callback (round 5): for every published person either the refined joints agree (< REFINE_TOL_M on
    every valid joint, same set of valid joints) or the library's final objective — graph.error(result) = 0.5 sum ||whitened r||^2, PP:746-758,
    se3d_fused_out.lm_objective — is not above the oracle's (x (1 + OBJECTIVE_RTOL)).  Both sides refine the SAME stage-1 persons, so the objective
    FUNCTION of a person is the same on both sides whatever the track state; it is comparable whenever both sides publish the same detections
    (equal counts, equal measured scores).  Returns dict(comparable, persons, disagree, worse, worst_excess)."""
    out = dict(comparable=False, persons=0, disagree=0, worse=0, worst_excess=0.0, pred_only=0)
    n = gpu.n_persons
    if n != ora.n_persons:
        return out
    out["comparable"] = True
    if n == 0:
        return out
    g, o = gpu.fused[:n]["keypoints"], ora.fused[:n]["keypoints"]
    if not np.array_equal(g["score"], o["score"]):   # not the same detections (the published subsets differ): nothing to compare
        out["comparable"] = False
        return out
    v = o["score"] > 0
    d = np.where(v, np.linalg.norm(g["joint"] - o["joint"], axis=-1), 0.0).max(-1)
    og, oo = gpu.lm_objective[:n], ora.lm_objective[:n]
    out["persons"] = int(n)
    # round 6: the PREDICTED joints (fused_pred = refined joint + velocity x predicted latency, PP:795-837).  A person whose refined joints agree but
    # whose predicted joints do not carries a different velocity history on the two sides (an earlier step's refined joints differed): the objective
    # says nothing about it — counted on its own (`pred_only`) and bounded by the callers
    gp, op_ = gpu.fused_pred[:n]["keypoints"], ora.fused_pred[:n]["keypoints"]
    vp = op_["score"] > 0
    dp = np.where(vp & (gp["score"] > 0), np.linalg.norm(gp["joint"] - op_["joint"], axis=-1), 0.0).max(-1)
    for p in range(n):
        if d[p] < REFINE_TOL_M:
            out["pred_only"] += int(dp[p] >= REFINE_TOL_M)
            continue
        out["disagree"] += 1
        if (og[p] < 0) != (oo[p] < 0) or og[p] > oo[p] * (1.0 + OBJECTIVE_RTOL):
            out["worse"] += 1
            out["worst_excess"] = max(out["worst_excess"], float(og[p] / oo[p] - 1.0) if oo[p] > 0 and og[p] >= 0 else float("inf"))
    return out


# ---------------------------------------------------------------------------------------------------------------------------------
# Conditioning of the DLT systems, computed FROM THE INPUTS in float64 (tests/test_gpu_fuzz_parity.py).
#
# Rows T/U: the reference takes the DLT null vector from a float32 JacobiSVD (S3D:456), the fast kernels from an fp64 solve of the same
# float32 rows.  A float32 perturbation of relative size eps = 2^-24 of the rows turns the null vector by ~ eps * s1 / (s3 - s4)
# (singular values of the 2n x 4 system) and dehomogenising v -> v[:3] / v[3] multiplies that by ~ (1 + |X|^2): a joint whose system has
#     kappa = (1 + |X|^2) * s1 / (s3 - s4)
# can come out of the two solvers up to ~ c * eps * kappa metres apart (measured c <= 5 over 3e5 joints, profiles/r3/README.md).
# Round 3 EXCLUDED the joints with kappa >= DLT_KAPPA_MAX from the position / score / covariance assertions.  Round 4: the library flags such
# joints itself (an in-kernel upper bound of kappa at 1e3, reprojection errors near the outlier threshold, wide covariances) and redoes them
# with the oracle's restatement of the reference's float32 arithmetic — positions and scores are asserted on EVERY valid joint, and what is left of kappa here is
#   (i) a check of the library's flag: own-hypothesis joints with kappa >= DLT_KAPPA_MAX must be in se3d_get_exact_list, and
#   (ii) the covariance tolerance of the joints that stay on the fp64 fast path (cov_tolerance).
# ---------------------------------------------------------------------------------------------------------------------------------
DLT_KAPPA_MAX = 2.0e3     # twice the library's SE3D_EXACT_KAPPA (its bound is 1.0 .. 2.1 x kappa): every joint at or above it must be listed


COV_NOISE_CAP = 1e-2       # the float32-noise term of cov_tolerance never allows more than 1 % (measured maxima: 5.6e-4 at 256 x 128, 5.4e-3 worst of the blind batches)


def cov_tolerance(kappa, cov):
    """Relative-Frobenius tolerance of a stage-1 covariance: COV_REL_FRO + 2 * 2^-24 * kappa / sigma, sigma = sqrt(mean diagonal)
    of the oracle's covariance `cov` [..., 6].  The unscented transform differences 4n + 1 sample solutions that lie ~sigma apart; the oracle
    (like the reference) solves every sample in float32, whose noise on a sample is c 2^-24 kappa metres (c: median 0.31, maximum 4.91 over
    3.1e5 joints, profiles/r3/cond_study.txt), so its covariance carries a relative error of about 2 c 2^-24 kappa / sigma — the fast path's
    fp64 sample solves have none of it (CPU study of 46 k joints, profiles/r3/cond_study.txt: the oracle's deviation from the exact unscented
    transform divided by 2^-24 kappa / sigma has median 0.27 and 99th percentile 1.45).  The term is below 1e-4 for the bulk of the joints
    (kappa ~ 20, sigma ~ 2 cm) and decides the rare tail: over 86 seeds x 40 scenes (9.5 M asserted joints) three joints exceeded the flat 1e-3
    for this reason — 1.46e-3 at kappa 62 / sigma 4 mm (19 views), 1.01e-3 at kappa 744, 1.2e-3 — all inside this tolerance.
    Callers pass min(kappa, DLT_KAPPA_MAX): joints above it went through the library's float32 second pass and carry the oracle's bits.
    The added term is capped at COV_NOISE_CAP (round 5: uncapped it allowed 24 % at kappa = 2e3, sigma = 1 mm — 40 x looser than anything measured)."""
    cov = np.asarray(cov, np.float64)
    with np.errstate(invalid="ignore"):
        sigma = np.sqrt(np.maximum((cov[..., 0] + cov[..., 3] + cov[..., 5]) / 3.0, 1e-300))
    return COV_REL_FRO + np.minimum(2.0 * 2.0 ** -24 * np.asarray(kappa, np.float64) / sigma, COV_NOISE_CAP)


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


def plain_condition(rig, frame, assoc_rows, want_points=False):
    """kappa [n][17]: maximum over the confidence-weighted (S3D:746) and the unweighted (S3D:518) DLT system of each (hypothesis, joint)
    (nan: fewer than two usable views) — the quantity the library's in-kernel flag bounds from above (csrc/se3d_stage1.hip se3d_kappa_flag).
    assoc_rows: [n][n_cam] original detection indices or -1.  want_points: also the fp64 solution of the weighted system [n][17][3]."""
    assoc_rows = np.asarray(assoc_rows, np.int32).reshape(-1, rig.n_cam)
    r0, r1, use, cf, x, y = _dlt_rows(rig, frame, assoc_rows)
    a0, a1 = r0.astype(np.float64), r1.astype(np.float64)
    outer = a0[..., :, None] * a0[..., None, :] + a1[..., :, None] * a1[..., None, :]
    Gu = (outer * use.astype(np.float64)[..., None, None]).sum(2)
    Gw = (outer * (cf.astype(np.float64) ** 2)[..., None, None]).sum(2)
    ku, _ = _kappa_from_gram(Gu)
    kw, X = _kappa_from_gram(Gw)
    kappa = np.where(use.sum(-1) >= 2, np.maximum(ku, kw), np.nan)
    return (kappa, X) if want_points else kappa


def published_joint_condition(rig, frame, result, k2f=None):
    """kappa [n_persons][17] for the PUBLISHED persons of a stage-1 result (keypoint order of the input, not the fusion layout).  A published
    person is its own hypothesis or, after the duplicate merge (S3D:984-996: the later of two persons closer than 0.20 m is averaged into
    the earlier one and erased, joints only one of them has are taken over), a blend: every unpublished later hypothesis whose fp64 joints
    lie within 0.5 m (median over the joints both have; the reference's test runs after its plausibility filters, which this criterion
    does not redo) or that has one joint within 2 cm (a joint taken over unchanged) counts as a possible merge partner and contributes the maximum of its kappa."""
    n = result.n_persons
    hyp = result.person_hyp[:n]
    kap, X = plain_condition(rig, frame, result.assoc(), want_points=True)
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
