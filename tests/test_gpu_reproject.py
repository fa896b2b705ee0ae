"""GPU parity of the feedback reprojection (C-ABI se3d_reproject / se3d_reproject_resident) against the CPU oracle.

Tolerances: membership / order of every per-camera list exact; keypoint validity exact; pixel means, bbox and 2x2 covariances are
fp64 on both sides and stored as float32 -> equal to 1 float32 ulp (2e-6 relative, 1e-4 px absolute)."""
import numpy as np
import pytest

from smartedgesensor3dhumanpose_amd import person_msgs as pm
from smartedgesensor3dhumanpose_amd.pose_reprojection import SkeletonReprojection
from smartedgesensor3dhumanpose_amd.rig import elliptical_rig, extr_rig_16cam
from smartedgesensor3dhumanpose_amd.scene import SceneConfig, generate_sequence
from smartedgesensor3dhumanpose_amd.skeleton_3d import Se3dError, Skeleton3D

from test_oracle_reproject import make_persons

pytestmark = pytest.mark.gpu


def compare(res, ref, n_cam):
    assert np.array_equal(res.det_offset, ref.det_offset)
    tot = int(ref.det_offset[-1])
    assert np.array_equal(res.person_index[:tot], ref.person_index[:tot])
    a, b = res.persons[:tot], ref.persons[:tot]
    assert np.array_equal(a["keypoints"]["score"], b["keypoints"]["score"])
    assert np.array_equal(a["score"], b["score"])
    for f in ("x", "y", "cov"):
        np.testing.assert_allclose(a["keypoints"][f], b["keypoints"][f], rtol=2e-6, atol=1e-4)
    np.testing.assert_allclose(a["bbox"], b["bbox"], rtol=2e-6, atol=1e-4)
    return tot


@pytest.mark.parametrize("method", ["simple", "h36m"])
def test_reproject_random_persons(oracle, method):
    rig = extr_rig_16cam()
    node = Skeleton3D(rig, pose_method=method, max_det_per_cam=8, max_hyp=32)
    node.set_reprojection_cameras()
    rng = np.random.default_rng(11)
    for n in (0, 1, 7, 32):
        persons = make_persons(rng, n, rig)
        res = node.reproject(persons)
        ref = oracle.reproject(rig, method, persons)
        tot = compare(res, ref, rig.n_cam)
        assert (tot > 0) == (n > 0)
    node.close()


def test_reproject_errors():
    rig = extr_rig_16cam()
    node = Skeleton3D(rig, max_det_per_cam=8, max_hyp=8)
    with pytest.raises(Se3dError):
        node.reproject(np.zeros(1, pm.person3d_dt))  # cameras not set
    node.set_reprojection_cameras()
    with pytest.raises(Se3dError):
        node.reproject(np.zeros(9, pm.person3d_dt))  # capacity
    node.close()


def test_reproject_callback_interface(oracle):
    rig = extr_rig_16cam()
    node = Skeleton3D(rig, max_det_per_cam=8, max_hyp=16)
    rp = SkeletonReprojection(node)
    persons = make_persons(np.random.default_rng(3), 5, rig)
    ts = [100.0 + 0.001 * i for i in range(rig.n_cam)]
    fb = [0.05] * rig.n_cam
    lists = rp.fused_skeleton_callback(persons, ts, fb)
    ref = oracle.reproject(rig, "simple", persons)
    assert len(lists) == rig.n_cam
    for i, l in enumerate(lists):
        assert l.stamp == ts[i] and l.fb_delay == fb[i]
        assert np.array_equal(l.person_index, ref.camera(i)[1])
    with pytest.raises(ValueError):
        rp.fused_skeleton_callback(persons, ts, fb, frame_id="map")
    node.close()


def test_reproject_resident_after_stage2(oracle):
    """skeleton_3d -> pose_prior -> reprojection without leaving HBM: the resident entry equals the oracle's reprojection of the
    fused_pred skeletons that se3d_fetch_stage2 returns."""
    rig = extr_rig_16cam()
    cfg = SceneConfig(n_persons=4, n_frames=16, seed=21)
    frames, _ = generate_sequence(rig, cfg)
    node = Skeleton3D(rig, pose_method="simple", max_epi_dist=0.045, max_frames=1, max_det_per_cam=8, max_hyp=16)
    node.set_reprojection_cameras()
    n_checked = 0
    for f in frames:
        node.stage([f])
        node.run_stage1()
        node.run_stage2([f.stamp])
        fused = node.fetch_stage2(1)[0]
        res = node.reproject_resident(0)
        pred = fused.fused_pred[:fused.n_persons]
        ref = oracle.reproject(rig, "simple", pred)
        n_checked += compare(res, ref, rig.n_cam)
    assert n_checked > 50
    node.close()


def test_reproject_resident_batch_after_walk(oracle):
    """The reprojection of a replayed batch in ONE launch (se3d_reproject_resident_batch over every frame slot of a resident walk): every
    slot's per-camera lists equal the single-slot entry byte for byte and the oracle's reprojection of the predicted skeletons of that
    slot; slots without published persons give empty lists."""
    rig = extr_rig_16cam()
    S_, T_ = 3, 14
    seqs = [generate_sequence(rig, SceneConfig(n_persons=3 + s, n_frames=T_, seed=40 + s))[0] for s in range(S_)]
    frames = [seqs[s][k] for k in range(T_) for s in range(S_)]
    node = Skeleton3D(rig, pose_method="simple", max_epi_dist=0.045, max_frames=S_ * T_, max_det_per_cam=8, max_hyp=12)
    node.set_reprojection_cameras()
    node.stage(frames)
    node.run_stage1()
    node.refine_reset()
    ts = [100.0 + (i + 1) / 30.0 for i in range(T_)]
    node.run_stage2_walk(np.repeat(np.array(ts)[:, None], S_, 1), [k * S_ for k in range(T_)])
    node.reproject_resident_batch(0, S_ * T_)
    n_checked = n_empty = 0
    for slot in range(S_ * T_):
        got = node.fetch_reprojection(slot)
        fused = node.fetch_stage2_walk(slot, 1)[0]
        pred = fused.fused_pred[:fused.n_persons]
        ref = oracle.reproject(rig, "simple", pred)
        if fused.n_persons == 0:
            assert got.det_offset[-1] == 0
            n_empty += 1
            continue
        n_checked += compare(got, ref, rig.n_cam)
    assert n_checked > 200 and n_empty >= S_ * 10    # nothing is published before a track's 11th observation
    for slot in (S_ * T_ - 1, S_ * (T_ - 2) + 1):    # single-slot entry: same bytes
        one = node.reproject_resident(slot)
        node.reproject_resident_batch(S_ * (T_ - 3), 3 * S_)   # a sub-range that contains the slot
        got = node.fetch_reprojection(slot)
        n = int(one.det_offset[-1])
        assert np.array_equal(one.det_offset, got.det_offset) and n > 0
        assert one.persons[:n].tobytes() == got.persons[:n].tobytes() and np.array_equal(one.person_index[:n], got.person_index[:n])
    node.close()


def test_reproject_large_rig(oracle):
    rig = elliptical_rig(64, seed=2)
    node = Skeleton3D(rig, max_det_per_cam=8, max_hyp=128)
    node.set_reprojection_cameras()
    persons = make_persons(np.random.default_rng(8), 128, rig, spread=6.0)
    res = node.reproject(persons)
    ref = oracle.reproject(rig, "simple", persons)
    assert compare(res, ref, rig.n_cam) > 1000
    node.close()
