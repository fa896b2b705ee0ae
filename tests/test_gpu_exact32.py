"""Round 4: the float32-exact second pass of the triangulation (csrc/se3d_exact32.h, se3d_k_triangulate_exact).

The reference takes every DLT null vector from Eigen's JacobiSVD in float (S3D:456).  Joints the fast fp64 kernels flag — ill-conditioned
systems (se3d_kappa_bound), reprojection errors near or above g_reproj_error_max_acceptable, unsettled solves — are redone with that
arithmetic.  IEEE float32 operations in a fixed order give the same bits on every machine, so on those joints the HIP path must equal the CPU
restatement of the reference BIT FOR BIT: positions, scores and decisions of the outlier rejection (S3D:748-844)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import branch_scenes  # noqa: E402
from smartedgesensor3dhumanpose_amd import person_msgs as pm, rig as R, scene as S  # noqa: E402
from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D, rig_tensors  # noqa: E402

pytestmark = pytest.mark.gpu


def _systems(rig, rng, n_sys, n, ill=False):
    """n_sys DLT systems of n views: a 3-D point seen by n cameras of the rig with pixel noise (ill: two nearly parallel rays + outliers)."""
    Rt = np.zeros((n_sys, n, 12), np.float32)
    pts = np.zeros((n_sys, n, 3), np.float32)
    M = rig.Rt.reshape(rig.n_cam, 3, 4).astype(np.float64)
    for s in range(n_sys):
        cams = rng.choice(rig.n_cam, n, replace=n > rig.n_cam)
        X = np.array([rng.uniform(-4, 2), rng.uniform(-3, 3), rng.uniform(0.1, 1.9), 1.0])
        if ill:
            X[:3] *= rng.choice([1.0, 5.0, 40.0])
        for i, c in enumerate(cams):
            h = M[c] @ X
            z = h[2] if abs(h[2]) > 1e-3 else 1e-3
            noise = rng.normal(0, 3e-3 * (10.0 if ill and rng.random() < 0.2 else 1.0), 2)
            pts[s, i] = [h[0] / z + noise[0], h[1] / z + noise[1], rng.uniform(0.3, 0.95)]
            Rt[s, i] = rig.Rt[c]
    return Rt, pts


@pytest.mark.parametrize("n", [2, 3, 4, 7, 16])
@pytest.mark.parametrize("weighted", [True, False])
def test_float32_dlt_solve_equals_the_oracle_bit_for_bit(oracle, extr_rig, n, weighted):
    rng = np.random.default_rng(100 + n)
    node = Skeleton3D(extr_rig, max_frames=1, max_det_per_cam=4, max_hyp=4, device=0)
    for ill in (False, True):
        Rt, pts = _systems(extr_rig, rng, 700, n, ill)
        X, err = node.selftest_dlt32(Rt, pts, weighted)
        for s in range(len(X)):
            Xo, eo = oracle.dlt(Rt[s], pts[s], weighted)
            assert np.array_equal(X[s].view(np.uint32), Xo.view(np.uint32)), (n, weighted, ill, s, X[s], Xo)
            if weighted:   # the oracle entry evaluates the error with the confidences as weights in both modes
                assert np.float64(err[s]).tobytes() == np.float64(eo).tobytes() or (np.isnan(err[s]) and np.isnan(eo)), (n, s, err[s], eo)
    node.close()


@pytest.mark.parametrize("n_cam,n", [(48, 40), (160, 130)])
def test_float32_dlt_solve_on_large_rigs(oracle, n_cam, n):
    rig = R.elliptical_rig(n_cam, seed=3)
    rig.set_tensors(*rig_tensors(rig))
    rng = np.random.default_rng(n)
    node = Skeleton3D(rig, max_frames=1, max_det_per_cam=4, max_hyp=4, device=0)
    Rt, pts = _systems(rig, rng, 128, n)
    X, err = node.selftest_dlt32(Rt, pts, True)
    for s in range(len(X)):
        Xo, eo = oracle.dlt(Rt[s], pts[s], True)
        assert np.array_equal(X[s].view(np.uint32), Xo.view(np.uint32)), (s, X[s], Xo)
        assert np.float64(err[s]).tobytes() == np.float64(eo).tobytes()
    node.close()


W6 = np.array([1, 2, 2, 1, 2, 1.0])
COV64 = {"listed": 0, "cov64": 0, "rel_max": 0.0}   # tally over the module's runs (printed by the last test)


def _listed_joints_equal_bitwise(oracle, rig, frames, pose_method="simple", n_hyp=32, max_det=8, exact_cov="auto"):
    """Config 2 (rows N, T, P, O, U with the association given: no plausibility filters, no duplicate merge, no limb-length inflation) on
    `frames` with the association the oracle's sweep finds: every joint that went through the second pass must carry the oracle's position
    and score bits, and its raw covariance bits too when its sigma samples were solved in float32 — every listed joint with
    exact_cov="f32" (the form of rounds 4 - 5), the joints whose kept views are ill conditioned / whose covariance is wide by default (round 6).
    The other listed joints (bit 31 of the list entry) take the fast path's fp64 samples around the float32 point: their covariance is held
    to the fast path's contract, parity.cov_tolerance with the input-derived kappa.  Returns (listed joints compared, joints triangulated)."""
    from parity import cov_tolerance, plain_condition, DLT_KAPPA_MAX
    node = Skeleton3D(rig, pose_method=pose_method, max_epi_dist=0.045, max_frames=len(frames), max_det_per_cam=max_det, max_hyp=n_hyp, device=0,
                      options=dict(exact_cov=exact_cov))
    k2f = np.array(oracle.pose_tables(pm.POSE_H36M if pose_method == "h36m" else pm.POSE_SIMPLE)["kp2fusion"])
    assoc = np.full((len(frames), n_hyp, rig.n_cam), -1, np.int32)
    for k, fr in enumerate(frames):
        a = oracle.triangulate(rig, node.params, fr, capacity=n_hyp, hyp_capacity=n_hyp).assoc()
        assoc[k, :len(a)] = a
    got = node.triangulate_assoc(frames, assoc)
    listed, cov64, _ = node.exact_list(flags=True)
    assert exact_cov == "auto" or not cov64.any(), "exact_cov = f32 must solve every listed joint's samples in float32"
    triangulated = 0
    for k, fr in enumerate(frames):
        ref = oracle.triangulate_assoc(rig, node.params, fr, assoc[k])
        gk, ok = got[k]["keypoints"][:, k2f], ref["keypoints"][:, k2f]
        assert np.array_equal(gk["score"] > 0, ok["score"] > 0)
        triangulated += int((ok["score"] > 0).sum())
        kappa = None
        sel = listed[:, 0] == k
        for (f_, h, j), c64 in zip(listed[sel], cov64[sel]):
            assert ok["score"][h, j] > 0, ("listed joint the oracle did not triangulate", k, h, j)
            assert np.array_equal(gk["joint"][h, j], ok["joint"][h, j]), ("position bits", k, h, j, gk["joint"][h, j], ok["joint"][h, j])
            assert gk["score"][h, j].tobytes() == ok["score"][h, j].tobytes(), ("score bits", k, h, j, gk["score"][h, j], ok["score"][h, j])
            cg, co = gk["cov"][h, j], ok["cov"][h, j]
            COV64["listed"] += 1
            if not c64:
                assert cg.tobytes() == co.tobytes() or (np.isnan(cg).all() and np.isnan(co).all()), ("covariance bits", k, h, j, cg, co)
                continue
            if kappa is None:
                kappa = plain_condition(rig, fr, assoc[k])
            rel = float(np.sqrt(((cg - co) ** 2 * W6).sum()) / np.sqrt((co ** 2 * W6).sum()))
            tol = float(cov_tolerance(min(kappa[h, j], DLT_KAPPA_MAX), co))
            COV64["cov64"] += 1
            COV64["rel_max"] = max(COV64["rel_max"], rel)
            assert rel < tol, ("fp64-sample covariance of a listed joint outside the fast path's tolerance", k, h, j, rel, tol, kappa[h, j])
        # everything else: the fast path's contract
        rest = ok["score"] > 0
        rest[tuple(listed[listed[:, 0] == k][:, 1:].T)] = False
        if rest.any():
            assert np.linalg.norm(gk["joint"] - ok["joint"], axis=-1)[rest].max() < 1e-3
    node.close()
    return len(listed), triangulated


def test_outlier_scenes_second_pass_is_bit_identical(oracle, extr_rig):
    """8 % gross 2-D outliers, with and without missing views: 3-view epipolar rejection, leave-one-out rejection and score down-weighting all
    run in the second pass and give the oracle's bits — positions, scores and unscented-transform covariances."""
    for p_miss, exact_cov in ((0.0, "auto"), (0.0, "f32"), (0.6, "auto"), (0.6, "f32")):
        oracle.stats_reset()
        frames = branch_scenes.outliers(extr_rig, n_frames=6, p_miss=p_miss)
        listed, triangulated = _listed_joints_equal_bitwise(oracle, extr_rig, frames, exact_cov=exact_cov)
        st = oracle.stats()
        print("p_miss %.1f: %d of %d triangulated joints went through the second pass, all bit-identical; oracle branches %s" % (
            p_miss, listed, triangulated, {k: st[k] for k in ("outlier3_removed", "outlier_loo_removed", "score_downweighted")}))
        assert listed > 5 and st["score_downweighted"] > 0
        assert st["outlier_loo_removed"] > 0 if p_miss == 0.0 else st["outlier3_removed"] > 0


@pytest.mark.parametrize("pose_method,exact_cov", [("simple", "auto"), ("simple", "f32"), ("h36m", "auto")])
def test_random_rough_scenes_second_pass_is_bit_identical(oracle, extr_rig, pose_method, exact_cov):
    """Scenes of the randomised sweep's kind (missing views, low confidences, gross outliers, close persons) on the 16-camera rig."""
    rng = np.random.default_rng(7)
    tot = 0
    for trial in range(6):
        cfg = S.SceneConfig(n_persons=int(rng.integers(2, 7)), n_frames=3, seed=int(rng.integers(1 << 30)), p_miss=float(rng.uniform(0, 0.5)),
                            low_score_frac=float(rng.uniform(0, 0.3)), outlier_frac=float(rng.uniform(0, 0.05)), pixel_noise=float(rng.uniform(0.5, 4.0)),
                            spacing=float(rng.uniform(0.5, 1.2)))
        frames = S.generate_sequence(extr_rig, cfg)[0]
        listed, triangulated = _listed_joints_equal_bitwise(oracle, extr_rig, frames, pose_method=pose_method, max_det=12, exact_cov=exact_cov)
        tot += listed
    print("%s / exact_cov %s: %d listed joints, all bit-identical (so far in this module: %s)" % (pose_method, exact_cov, tot, COV64))
    assert tot > 20


@pytest.mark.parametrize("n_cam,p_miss,exact_cov", [(32, 0.3, "auto"), (96, 0.0, "auto"), (96, 0.0, "f32")])
def test_large_rig_second_pass_is_bit_identical(oracle, n_cam, p_miss, exact_cov):
    """32 / 96 cameras: the per-joint fast kernel hands its flagged joints to the second pass; at 96 cameras the camera matrices come from
    global memory, the centred samples live in the workspace, and joints seen by more than 62 views take the leave-one-out passes without
    speculation (gross outliers on 5 % of the keypoints force them)."""
    rig = R.elliptical_rig(n_cam, seed=5)
    rig.set_tensors(*rig_tensors(rig))
    frames = S.generate_sequence(rig, S.SceneConfig(n_persons=5, n_frames=2, seed=3, outlier_frac=0.05, p_miss=p_miss))[0]
    oracle.stats_reset()
    listed, triangulated = _listed_joints_equal_bitwise(oracle, rig, frames, n_hyp=48, max_det=12, exact_cov=exact_cov)
    st = oracle.stats()
    print("%d cameras: %d of %d joints listed, all bit-identical; oracle branches %s" % (n_cam, listed, triangulated, {k: st[k] for k in ("outlier_loo_removed", "score_downweighted")}))
    assert listed > 10 and st["outlier_loo_removed"] > 0


@pytest.mark.parametrize("n_cam", [5, 9])
def test_small_rig_second_pass_is_bit_identical(oracle, n_cam):
    """Rigs with FEWER than 16 cameras (round 6): the four-joints-per-wavefront kernel runs with lanes beyond the camera count idle and short view lists
    (three-view epipolar rejection and leave-one-out removals are frequent here: gross outliers on 8 % of the keypoints)."""
    rig = R.elliptical_rig(n_cam, seed=5)
    rig.set_tensors(*rig_tensors(rig))
    frames = S.generate_sequence(rig, S.SceneConfig(n_persons=4, n_frames=3, seed=3, outlier_frac=0.08, min_coverage=min(4, n_cam - 1)))[0]
    oracle.stats_reset()
    listed, triangulated = _listed_joints_equal_bitwise(oracle, rig, frames, n_hyp=48, max_det=12)
    st = oracle.stats()
    print("%d cameras: %d of %d joints listed, all bit-identical; oracle branches %s" % (n_cam, listed, triangulated, {k: st[k] for k in ("outlier3_removed", "outlier_loo_removed", "score_downweighted")}))
    assert listed > 5 and st["outlier3_removed"] + st["outlier_loo_removed"] > 0


def test_clean_scene_stays_on_the_fast_path(oracle, extr_rig):
    """The benchmark's kind of scene (no gross outliers, all views): a fraction of a percent is listed — the second pass costs the headline little."""
    frames, _ = S.generate_sequence(extr_rig, S.SceneConfig(n_persons=6, n_frames=8, seed=0))
    listed, triangulated = _listed_joints_equal_bitwise(oracle, extr_rig, frames)
    print("clean scene: %d of %d triangulated joints listed; module tally of the covariance forms: %s" % (listed, triangulated, COV64))
    assert listed <= 0.01 * triangulated
    assert COV64["cov64"] > 0 and COV64["cov64"] < COV64["listed"], "both covariance forms of the second pass must have been exercised"
