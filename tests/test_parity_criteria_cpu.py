"""CPU checks of the parity criteria themselves (tests/parity.py): they are test infrastructure, but the GPU sweep's verdict depends on them.
Oracle results stand in for both sides (no GPU)."""
import copy
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import parity  # noqa: E402
from smartedgesensor3dhumanpose_amd import person_msgs as pm  # noqa: E402
from smartedgesensor3dhumanpose_amd import scene as S  # noqa: E402


def _frame_and_result(extr_rig, oracle, seed=5, **kw):
    frames, _ = S.generate_sequence(extr_rig, S.SceneConfig(n_persons=5, n_frames=1, seed=seed, **kw))
    res = oracle.triangulate(extr_rig, pm.Params(pm.POSE_SIMPLE, 0, 0.045), frames[0], capacity=32, hyp_capacity=32)
    return frames[0], res


def test_allowances_vanish_for_identical_sides_and_grow_with_the_difference(extr_rig, oracle):
    fr, o = _frame_and_result(extr_rig, oracle, outlier_frac=0.05, pixel_noise=3.0)
    assert o.n_persons >= 4
    assert np.array_equal(parity.score_allowance(extr_rig, fr, o, o), np.zeros((o.n_persons, pm.NUM_KEYPOINTS)))
    kap = parity.published_joint_condition(extr_rig, fr, o)
    v = o.published()["keypoints"]["score"][:, np.array(pm.KP2FUSION_SIMPLE)] > 0
    assert not np.isnan(kap[v]).any() and (kap[v] > 1.0).all()
    tables = oracle.pose_tables(pm.POSE_SIMPLE)
    kc = parity.covariance_condition(kap, tables)
    with np.errstate(invalid="ignore"):
        assert (np.nan_to_num(kc, nan=0.0) >= np.nan_to_num(kap, nan=0.0)).all()          # the parent's system can only add
    # a side whose points are 0.5 mm off: the score allowance is non-zero exactly where the reprojection error comes near / above the threshold
    o2 = copy.copy(o)
    o2.persons = o.persons.copy()
    o2.persons["keypoints"]["joint"][: o.n_persons] += 5e-4
    sa = parity.score_allowance(extr_rig, fr, o, o2)
    assert np.isfinite(sa).all() and (sa >= 0).all() and (sa > 0).any() and (sa == 0).any()
    # covariance tolerance: the flat 1e-3 plus a term that grows with kappa and shrinks with sigma
    cov = np.zeros(6)
    cov[[0, 3, 5]] = 0.02 ** 2
    t = parity.cov_tolerance(np.array([10.0, 100.0, 1000.0]), cov)
    assert t[0] > parity.COV_REL_FRO and np.all(np.diff(t) > 0)
    cov[[0, 3, 5]] = 0.004 ** 2
    assert parity.cov_tolerance(62.0, cov) > 1.46e-3                                            # the 19-view joint of seed 320
    assert parity.covariance_in_hall(cov) and not parity.covariance_in_hall(np.array([1157.0, -1056.0, 404.0, 963.0, -369.0, 141.0]))


def test_garbage_persons_flags_distance_and_sigma():
    p = np.zeros(3, pm.person3d_dt)
    p["keypoints"]["score"][:, :10] = 0.7
    p["keypoints"]["joint"][:, :10] = np.random.default_rng(0).normal(0, 0.3, (3, 10, 3))
    p["keypoints"]["cov"][:, :10, [0, 3, 5]] = 0.02 ** 2
    assert not parity.garbage_persons(p).any()
    p["keypoints"]["joint"][1, 4] += 50.0                                  # a joint 50 m from the body
    p["keypoints"]["cov"][2, 7, [0, 3, 5]] = 761.0 ** 2                    # a joint near the body that is unknown to 761 m
    assert parity.garbage_persons(p).tolist() == [False, True, True]
    p["keypoints"]["cov"][2, 7, [0, 3, 5]] = 50.0 ** 2                     # 50 m: still a usable (weak) unary factor
    assert parity.garbage_persons(p).tolist() == [False, True, False]


def test_branch_tie_of_the_outlier_rejection_is_in_the_exclusion_set(extr_rig, oracle):
    """A view is moved until the fp64 reprojection error of the weighted solution sits on g_reproj_error_max_acceptable: the criterion must
    return inf for that joint (the two sides decide the rejection differently there), and a finite kappa a little away from the threshold."""
    fr, o = _frame_and_result(extr_rig, oracle, seed=9)
    h = int(o.person_hyp[0])
    row = o.assoc()[h]
    cams = np.nonzero(row >= 0)[0]
    assert len(cams) >= 5
    c, j = int(cams[0]), 5
    det = fr.det_offset[c] + row[c]

    def err_with_offset(dx):
        f2 = copy.copy(fr)
        f2.persons = fr.persons.copy()
        f2.persons["keypoints"]["x"][det, j] += dx
        _, _, use, cf, x, y = parity._dlt_rows(extr_rig, f2, o.assoc()[h:h + 1])
        kap, X = parity.dlt_condition(extr_rig, f2, o.assoc()[h:h + 1], want_points=True)
        P = extr_rig.Rt.reshape(extr_rig.n_cam, 3, 4).astype(np.float64)
        pr = P @ np.append(X[0, j], 1.0)
        e = np.hypot(pr[:, 0] / pr[:, 2] - x[0, j], pr[:, 1] / pr[:, 2] - y[0, j])
        w = np.where(use[0, j], cf[0, j].astype(np.float64), 0.0)
        return float((w * np.where(use[0, j], e, 0.0)).sum() / w.sum()), float(kap[0, j])
    lo, hi = 0.0, 400.0                                                     # pixels: bisect the offset that puts the error on the threshold
    assert err_with_offset(lo)[0] < 0.05 < err_with_offset(hi)[0]
    for _ in range(60):
        mid = 0.5 * (lo + hi)
        if err_with_offset(mid)[0] < 0.05:
            lo = mid
        else:
            hi = mid
    e_tie, k_tie = err_with_offset(hi)
    if abs(e_tie - 0.05) <= parity.BRANCH_TIE_REL * 0.05:                   # (the float32 pixel grid may not reach the band: then only the far side is checked)
        assert k_tie == np.inf
    e_far, k_far = err_with_offset(hi * 1.05)
    assert abs(e_far - 0.05) > parity.BRANCH_TIE_REL * 0.05 and np.isfinite(k_far)
