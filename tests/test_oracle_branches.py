"""CPU: every branch scene of tests/branch_scenes.py makes the ORACLE take the branch it is named after (counter > 0) and the
result shows the branch's effect.  The GPU twin (tests/test_gpu_branches.py) then proves the HIP path takes it too."""

import numpy as np

import branch_scenes as BS
from smartedgesensor3dhumanpose_amd import person_msgs as pm


def _s1(oracle, rig, params, frames):
    oracle.stats_reset()
    res = [oracle.triangulate(rig, params, fr, capacity=64, hyp_capacity=64) for fr in frames]
    return res, oracle.stats()


def test_duplicate_merge_fires(oracle, extr_rig, demo_params):
    res, st = _s1(oracle, extr_rig, demo_params, BS.duplicate_merge(extr_rig))
    assert st["duplicate_merge"] >= len(res)
    for r in res:
        assert r.n_hyp == 4 and r.n_persons == 3   # four people in the scene, two of them merged into one


def test_root_distance_reset_fires(oracle, extr_rig, demo_params):
    res, st = _s1(oracle, extr_rig, demo_params, BS.root_distance_reset(extr_rig))
    assert st["root_dist_reset"] >= len(res)
    lw = pm.FUSION["LWrist"]
    for r in res:
        assert r.n_persons == 3
        assert (r.published()["keypoints"]["score"][:, lw] == 0).sum() == 1   # exactly one person lost its left wrist


def test_feet_height_reject_fires(oracle, extr_rig, demo_params):
    res, st = _s1(oracle, extr_rig, demo_params, BS.feet_height_reject(extr_rig))
    assert st["feet_reject"] >= len(res) and st["person_rejected"] >= len(res)
    assert all(r.n_hyp == 3 and r.n_persons == 2 for r in res)


def test_outlier_branches_fire(oracle, extr_rig, demo_params):
    _, st = _s1(oracle, extr_rig, demo_params, BS.outliers(extr_rig) + BS.outliers(extr_rig, p_miss=0.6))
    for k in ("outlier3_removed", "outlier_loo_removed", "score_downweighted", "spawn_unmasked", "sweep_hungarian"):
        assert st[k] > 0, k


def test_non_spd_cov2d_gives_nan_covariance_only(oracle, extr_rig, demo_params):
    res, st = _s1(oracle, extr_rig, demo_params, BS.non_spd_cov2d(extr_rig))
    assert st["nan_covariance"] > 0
    ls = pm.FUSION["LShoulder"]
    for r in res:
        kp = r.published()["keypoints"]
        nan = np.isnan(kp["cov"]).any(-1)
        assert nan.any() and np.isnan(kp["cov"][nan]).all()
        assert set(np.nonzero(nan)[1]) == {ls}                     # only the joint that carried the indefinite covariance
        assert np.isfinite(kp["joint"]).all() and (kp["score"][nan] > 0).all()   # position and score unaffected


def test_track_merge_and_id_rewrite_fire(oracle, extr_rig, demo_params):
    seq = BS.converging_tracks(extr_rig, lambda fr: oracle.triangulate(extr_rig, demo_params, fr, capacity=32, hyp_capacity=32).published().copy())
    pp = oracle.PosePrior(demo_params)
    oracle.stats_reset()
    ids = []
    for fr, dets in seq:
        r = pp.step(dets, extr_rig.n_cam, fr.stamp, fr.fb_delay, capacity=16)
        ids.append([int(i) for i in r.fused["id"][: r.n_persons]])
    st = oracle.stats()
    assert st["track_merge"] > 0 and st["id_rewrite"] > 0
    assert [0, 0, 2] in ids     # the frame of the merge publishes the clone under the surviving track's id
    assert ids[-1] == [0, 2]


def test_lm_failure_publishes_measurements_with_default_covariance(oracle, extr_rig, demo_params):
    frames = BS.duplicate_merge(extr_rig, n_frames=12)   # any 12-frame sequence
    pp = oracle.PosePrior(demo_params)
    le = pm.FUSION["LElbow"]
    for k, fr in enumerate(frames):
        dets = oracle.triangulate(extr_rig, demo_params, fr, capacity=32, hyp_capacity=32).published().copy()
        if k == len(frames) - 1:
            dets = BS.non_spd_cov3d(dets)
            oracle.stats_reset()
        r = pp.step(dets, extr_rig.n_cam, fr.stamp, fr.fb_delay, capacity=16)
    st = oracle.stats()
    assert st["lm_failed"] == 1 and st["marginals_failed"] == 1
    f = r.fused[0]["keypoints"]
    meas = dets[0]["keypoints"]
    assert r.n_persons >= 1 and r.lm_iterations[0] == 0
    assert np.allclose(f["joint"][le], meas["joint"][le], atol=1e-12)           # the measurement itself (PP:751-758)
    assert np.allclose(f["cov"][le], [0.01, 0, 0, 0.01, 0, 0.01])               # g_defaultResSigma^2 I (PP:788-793)


def test_bone_factors_always_form_a_forest():
    """SE3D_S2FLAG_NOT_A_FOREST cannot be raised by any input: the range factors of addBinaryFactors (PP:384-481) connect measured
    joints only, and both maximal bone sets (Belly measured: no MidHip-Neck bone; Belly not measured: no bone through it) are forests,
    hence every subset of measured joints gives a forest."""
    bones = [(8, 9), (8, 12), (9, 10), (10, 11), (12, 13), (13, 14), (1, 0), (1, 2), (1, 5), (2, 3), (3, 4), (5, 6), (6, 7), (8, 20), (20, 1),
             (0, 19), (8, 1), (0, 15), (0, 16), (15, 17), (16, 18)]
    for belly in (True, False):
        edges = [(a, b) for a, b in bones if (belly and (a, b) != (8, 1)) or (not belly and 20 not in (a, b))]
        parent = list(range(21))

        def find(x):
            while parent[x] != x:
                x = parent[x]
            return x
        for a, b in edges:
            ra, rb = find(a), find(b)
            assert ra != rb, "cycle through bone %s" % ((a, b),)
            parent[ra] = rb
