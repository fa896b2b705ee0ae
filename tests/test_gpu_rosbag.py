"""GPU (-m gpu): §8f rank 3 end to end — a rosbag v2.0 file of per-camera person_msgs/Person2DList topics (written here from a synthetic
scene: the reference's demo bag is an external download) is read with rosbag_io, pushed through the approximate-time synchroniser with the
node's parameters (S3D:1218-1222) and the skeletonThreadCallback bookkeeping (pivot, stale views), and the synchronised tuples go through the
HIP path (stage 1 + stage 2).  The published lists must equal those of the same scene fed to the HIP path directly, and the oracle's."""
import os
import sys

import numpy as np
import pytest

from smartedgesensor3dhumanpose_amd import harness as Hn
from smartedgesensor3dhumanpose_amd import rosbag_io as rb
from smartedgesensor3dhumanpose_amd import scene as S

sys.path.insert(0, os.path.dirname(__file__))
from parity import compare_stage1, compare_stage2  # noqa: E402

pytestmark = pytest.mark.gpu


def test_bag_through_synchroniser_into_hip_path(tmp_path, extr_rig, oracle, demo_params):
    from smartedgesensor3dhumanpose_amd.pose_prior import PosePrior
    from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
    n_frames = 16
    frames, _ = S.generate_sequence(extr_rig, S.SceneConfig(n_persons=5, n_frames=n_frames, seed=41, p_miss=0.05))
    rng = np.random.default_rng(7)
    msgs = []
    for k, cams in enumerate(Hn.stream_from_frames(frames)):   # one message per camera and frame, stamps jittered by +-2 ms inside the frame
        for c, m in enumerate(cams):
            st = int(round(m.stamp * 1e9)) + int(rng.integers(-2_000_000, 2_000_000))
            msgs.append(("/cam_%d/human_joints" % c, "person_msgs/Person2DList", st + 5_000_000, rb.serialize_person2d_list(st, m.persons, m.fb_delay, "cam_%d" % c, k)))
    msgs.sort(key=lambda x: x[2])
    path = str(tmp_path / "poses2D_16cam.bag")
    rb.write_bag(path, msgs, chunk_messages=50)
    # ---- the bag is the only input from here on ----
    streams = rb.read_person2d_stream(path)
    topics = ["/cam_%d/human_joints" % c for c in range(16)]
    assert sorted(streams) == sorted(topics)
    tuples = []
    sync = Hn.ApproximateTimeSync(max(3, 1 + 16 // 4), 16, lambda tup: tuples.append([m for _, m in tup]), age_penalty=2.0, lower_bound_ns=20_000_000)
    for st, c, m in sorted((int(round(m.stamp * 1e9)), c, m) for c, t in enumerate(topics) for m in streams[t]):
        sync.add((st, m), c)
    assert len(tuples) >= n_frames - 2
    node = Skeleton3D(extr_rig, max_epi_dist=0.045, max_frames=1, max_det_per_cam=8, max_hyp=16)
    node.refine_reset()
    pp = PosePrior(node, 0)
    ora_pp = oracle.PosePrior(demo_params)
    stage1 = {}
    h1 = Hn.Skeleton3DHarness(16, lambda fr: stage1.setdefault("last", (fr, node.triangulate_persons([fr])[0]))[1])
    published = 0
    for k, tup in enumerate(tuples):
        stage1.clear()
        p3d = h1.step(tup)
        assert p3d is not None
        fr, got1 = stage1["last"]
        # the synchronised tuple is frame k of the scene: same detections as the generator produced
        ref = frames[k]
        assert fr.persons.tobytes() == ref.persons.tobytes() and np.array_equal(fr.det_offset, ref.det_offset)
        compare_stage1(got1, oracle.triangulate(extr_rig, demo_params, fr, capacity=16, hyp_capacity=16))
        g2 = pp.skeleton_callback(p3d.persons, p3d.stamp, np.array(p3d.fb_delay_per_cam, np.float32))
        o2 = ora_pp.step(p3d.persons, 16, p3d.stamp, np.array(p3d.fb_delay_per_cam, np.float32), capacity=16)
        compare_stage2(g2, o2)
        published += g2.n_persons
    assert published >= 3 * 5   # tracks publish from their 11th observation on
    assert h1.log == ["large frame delay 100.002"]   # the first frame after start-up (last_stamp 0); no stale views, no dropped frames
    node.close()
