"""BASELINE config 1 substitute (CPU only): a seeded synthetic 16-camera stream on the real rig pushed through the ROS-free
harness with the demo launch parameters (pose_method=simple, max_epi_dist=0.045, norm_height=false), CPU reference path
(= oracle backends).  Checks the host bookkeeping of skeletonThreadCallback (S3D:1029-1084)."""
import numpy as np

from smartedgesensor3dhumanpose_amd import harness as Hn
from smartedgesensor3dhumanpose_amd import person_msgs as pm
from smartedgesensor3dhumanpose_amd import scene as S


def _backends(oracle, rig, params):
    pp = oracle.PosePrior(params)
    tri = lambda fr: oracle.triangulate(rig, params, fr, capacity=64, hyp_capacity=64)
    ref = lambda persons, n_cam, t, fb: pp.step(persons, n_cam, t, fb, capacity=64)
    return tri, ref


def test_demo_stream_through_harness(oracle, extr_rig, demo_params):
    frames, gt = S.generate_sequence(extr_rig, S.SceneConfig(n_persons=6, n_frames=16, seed=0))
    stream = Hn.stream_from_frames(frames)
    tri, ref = _backends(oracle, extr_rig, demo_params)
    h1, h2 = Hn.Skeleton3DHarness(16, tri), Hn.PosePriorHarness(ref)
    n_pub = 0
    for k, msgs in enumerate(stream):
        p3d = h1.step(msgs)
        assert p3d is not None and len(p3d.persons) == 6
        assert p3d.ts_per_cam == [msgs[0].stamp] * 16 and all(abs(v + 1.0) < 1e-7 for v in p3d.fb_delay_per_cam)
        fused, pred = h2.callback(p3d)
        n_pub += len(fused.persons)
        assert all(abs(v - 0.10) < 1e-6 for v in fused.fb_delay_per_cam)  # predicted latency replaces fb_delay_per_cam (PP:531)
    assert n_pub == 6 * 6 and h1.log == ["large frame delay 100.000"]  # only the first frame (last_stamp starts at 0, S3D:1010)


def test_stale_view_replacement_and_non_monotonic_drop(oracle, extr_rig, demo_params):
    frames, _ = S.generate_sequence(extr_rig, S.SceneConfig(n_persons=3, n_frames=4, seed=3))
    stream = Hn.stream_from_frames(frames)
    tri, _ = _backends(oracle, extr_rig, demo_params)
    h = Hn.Skeleton3DHarness(16, tri)
    full = h.step(stream[0])
    # camera 0 lags 100 ms behind the pivot in frame 1 -> replaced by an empty dummy, its stamp / fb_delay still reported
    stream[1][0].stamp -= 0.100
    r = h.step(stream[1])
    assert "camera 0 stale, removed" in h.log and r.ts_per_cam[0] == stream[1][0].stamp
    ref_frame = pm.Frame(16, np.concatenate([[0], np.cumsum([0] + [len(m.persons) for m in stream[1][1:]])]).astype(np.int32),
                         np.concatenate([m.persons for m in stream[1][1:]]), None, stream[1][1].stamp)
    ref = oracle.triangulate(extr_rig, demo_params, ref_frame, capacity=64, hyp_capacity=64)
    assert r.persons.tobytes() == ref.published().tobytes()
    # a frame that does not advance time is dropped (S3D:1043-1046)
    assert h.step(stream[1]) is None and "non-monotonic frame dropped" in h.log
    # an 67 ms lag is still accepted (strict '>', S3D:1051)
    stream[2][5].stamp -= 0.060
    n_log = len(h.log)
    assert h.step(stream[2]) is not None and len(h.log) == n_log
    assert full is not None
