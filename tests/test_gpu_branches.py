"""GPU (-m gpu): the scenes of tests/branch_scenes.py through the C-ABI against the oracle.  Each test asserts (a) that the ORACLE took the
branch the scene is built for (branch counters, oracle_py.stats()) and (b) that the HIP path agrees with the oracle on that input —
association / ids / counts exactly, joints < 1 mm — i.e. it took the same branch.  Branch list: VERDICT r1 "What's weak" 2.

Branches of the reference that NO input can reach in this library, and why (documented, not tested on the GPU):
  * SE3D_S2FLAG_NOT_A_FOREST — the bone table never closes a cycle (tests/test_oracle_branches.py::test_bone_factors_always_form_a_forest);
  * SE3D_(S2)FLAG_ASSIGNMENT_FAILED — the Munkres iteration guard needs non-finite costs; finite joints give finite costs;
  * indeterminate marginals WITHOUT a failed optimisation (PP:788-793 alone) — every variable carries a full-rank unary factor, so J^T J is
    positive definite whenever the whitening succeeded; the default-covariance branch fires together with PP:751-758 (LM failure test).
"""
import os
import sys

import numpy as np
import pytest

from smartedgesensor3dhumanpose_amd import person_msgs as pm
from smartedgesensor3dhumanpose_amd import scene as S

sys.path.insert(0, os.path.dirname(__file__))
import branch_scenes as BS  # noqa: E402
from parity import compare_stage1, compare_stage2  # noqa: E402

pytestmark = pytest.mark.gpu


def _node(rig, **kw):
    from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
    kw.setdefault("max_epi_dist", 0.045)
    return Skeleton3D(rig, **kw)


STAGE1_SCENES = {
    # name: (builder, oracle counters that must fire, max_hyp values to run: <= 16 = one-wavefront sweep + post-processing kernels)
    "duplicate_merge": (lambda rig: BS.duplicate_merge(rig), ["duplicate_merge"], (8, 32)),
    "root_distance_reset": (lambda rig: BS.root_distance_reset(rig), ["root_dist_reset"], (8, 32)),
    "feet_height_reject": (lambda rig: BS.feet_height_reject(rig), ["feet_reject", "person_rejected"], (8, 32)),
    "outliers": (lambda rig: BS.outliers(rig) + BS.outliers(rig, p_miss=0.6),
                 ["outlier3_removed", "outlier_loo_removed", "score_downweighted", "spawn_unmasked", "sweep_hungarian"], (32,)),
    "non_spd_cov2d": (lambda rig: BS.non_spd_cov2d(rig), ["nan_covariance"], (8, 32)),
}


@pytest.mark.parametrize("tri", ["hyp", "joint"])
@pytest.mark.parametrize("name", sorted(STAGE1_SCENES))
def test_stage1_branch(extr_rig, oracle, demo_params, name, tri):
    build, counters, hyps = STAGE1_SCENES[name]
    frames = build(extr_rig)
    oracle.stats_reset()
    ora = [oracle.triangulate(extr_rig, demo_params, fr, capacity=64, hyp_capacity=64) for fr in frames]
    st = oracle.stats()
    for k in counters:
        assert st[k] > 0, "scene %s did not make the oracle take branch %s" % (name, k)
    for mh in hyps:   # tri = joint: one wavefront per (hypothesis, joint) instead of per third of a hypothesis
        node = _node(extr_rig, max_frames=len(frames), max_det_per_cam=8, max_hyp=mh, options=dict(tri_form="joint" if tri == "joint" else "auto"))
        got = node.triangulate_persons(frames)
        worst = 0.0
        for g, o in zip(got, ora):
            s = compare_stage1(g, o)
            worst = max(worst, s["pos_max_mm"])
            if name == "non_spd_cov2d":
                assert s["n_nan_cov"] > 0
        node.close()
        print("stage-1 branch %s (max_hyp %d, tri %s): max joint error %.5f mm" % (name, mh, tri, worst))


def test_stage2_track_merge_and_id_rewrite_host_path(extr_rig, oracle, demo_params):
    """PersonCov lists with a clone that closes in on its original (se3d_refine = one skeletonCallback per call)."""
    from smartedgesensor3dhumanpose_amd.pose_prior import PosePrior
    node = _node(extr_rig, max_frames=1, max_det_per_cam=8, max_hyp=16)
    seq = BS.converging_tracks(extr_rig, lambda fr: node.triangulate_persons([fr])[0].published().copy())
    node.refine_reset()
    pp, ora = PosePrior(node, 0), oracle.PosePrior(demo_params)
    oracle.stats_reset()
    ids = []
    for fr, dets in seq:
        g = pp.skeleton_callback(dets, fr.stamp, fr.fb_delay)
        o = ora.step(dets, extr_rig.n_cam, fr.stamp, fr.fb_delay, capacity=16)
        compare_stage2(g, o)
        ids.append([int(i) for i in g.fused["id"][: g.n_persons]])
    st = oracle.stats()
    assert st["track_merge"] > 0 and st["id_rewrite"] > 0
    assert [0, 0, 2] in ids and ids[-1] == [0, 2]
    node.close()


@pytest.mark.parametrize("variant", ["steps", "solo", "trio_wg", "trio_wave"])
def test_stage2_track_merge_resident(extr_rig, oracle, demo_params, variant):
    """Two people who slowly walk into each other, from 2-D detections through the RESIDENT pipeline: stage 1 on the GPU, then stage 2 as
    per-step launches or as the fused walk (three kernel forms).  Around the 0.20 m threshold stage 1 still reports two persons in some
    frames while their tracks are already closer than 0.20 m -> track merge + id rewrite inside the walk (PP:870-903)."""
    T_ = 40
    pos = S.generate_sequence(extr_rig, S.SceneConfig(n_persons=3, n_frames=1, seed=BS.SEED))[1]["positions"]
    cfgs = [S.SceneConfig(n_persons=3, n_frames=T_, seed=seed, positions=np.vstack([pos[:2], pos[0] + [0.245, 0.0]]), yaws=[0.3, 1.0, 0.3],
                          velocities=np.array([[0, 0], [0, 0], [-0.06, 0.0]]), joint_jitter=0.002) for seed in (12, 13, 14)]
    seqs = [S.generate_sequence(extr_rig, c)[0] for c in cfgs]
    S_ = len(seqs)
    frames = [seqs[s][k] for k in range(T_) for s in range(S_)]
    node = _node(extr_rig, max_frames=S_ * T_, max_det_per_cam=8, max_hyp=8)
    node.stage(frames)
    node.run_stage1()
    r1 = node.fetch_stage1(S_ * T_)
    node.refine_reset()
    ts = 100.0 + (np.arange(T_) + 1) / 30.0
    node.set_options(**{"solo": dict(walk_lm="solo"), "trio_wg": dict(walk_shape="wg"), "trio_wave": dict(walk_shape="wave")}.get(variant, {}))
    if variant != "steps":
        node.run_stage2_walk(np.repeat(ts[:, None], S_, 1), [k * S_ for k in range(T_)])
        node.sync()
    oras = [oracle.PosePrior(demo_params) for _ in range(S_)]
    oracle.stats_reset()
    for k in range(T_):
        if variant == "steps":
            node.run_stage2(np.full(S_, ts[k]), frame_slot0=k * S_)
            got = node.fetch_stage2(S_)
        else:
            got = node.fetch_stage2_walk(k * S_, S_)
        for s in range(S_):
            o = oras[s].step(r1[k * S_ + s].published(), 16, ts[k], seqs[s][k].fb_delay, capacity=8)
            compare_stage2(got[s], o, check_tracks=(variant == "steps" or k == T_ - 1))
    st = oracle.stats()
    assert st["track_merge"] >= 3 and st["id_rewrite"] >= 3, st
    node.close()


@pytest.mark.parametrize("lm", ["tree", "tree_mfma", "dense"])
def test_stage2_lm_failure_publishes_measurements(extr_rig, oracle, demo_params, lm):
    """An indefinite 3x3 covariance on one joint: the whitening of its unary factor fails = GTSAM's optimisation throws -> the person is
    published at its MEASUREMENTS with the default 0.10^2 covariance (PP:751-758, 788-793), zero LM iterations; the other persons of
    the frame are refined as usual."""
    from smartedgesensor3dhumanpose_amd.pose_prior import PosePrior
    frames = S.generate_sequence(extr_rig, S.SceneConfig(n_persons=3, n_frames=12, seed=5))[0]
    node = _node(extr_rig, max_frames=1, max_det_per_cam=8, max_hyp=16, options=dict(lm_form=lm))   # tree: the one-launch walk; the other forms exist in the per-step LM kernel only
    node.refine_reset()
    pp, ora = PosePrior(node, 0), oracle.PosePrior(demo_params)
    le = pm.FUSION["LElbow"]
    for k, fr in enumerate(frames):
        dets = node.triangulate_persons([fr])[0].published().copy()
        if k == len(frames) - 1:
            dets = BS.non_spd_cov3d(dets)
            oracle.stats_reset()
        g = pp.skeleton_callback(dets, fr.stamp, fr.fb_delay)
        o = ora.step(dets, extr_rig.n_cam, fr.stamp, fr.fb_delay, capacity=16)
        compare_stage2(g, o)
    st = oracle.stats()
    assert st["lm_failed"] == 1 and st["marginals_failed"] == 1
    assert g.n_persons == 3 and g.lm_iterations[0] == 0 and (g.lm_iterations[1:3] > 0).all()
    f = g.fused[0]["keypoints"]
    assert np.allclose(f["joint"][le], dets[0]["keypoints"]["joint"][le], atol=1e-12)
    assert np.allclose(f["cov"][le], [0.01, 0, 0, 0.01, 0, 0.01])
    node.close()
