"""Host-side mirror of the reference's pose_prior node interface (pose_prior_mult_node.cpp) over the HIP C-ABI.

`PosePrior.skeleton_callback(persons, t)` corresponds to skeletonCallback(PersonCovList) (PP:505): it consumes the stage-1
persons of one frame and returns the persons3d_fused / persons3d_fused_pred payloads (PP:905-907).  One PosePrior = one
sequence slot of a Skeleton3D context (the global state g_tracks / g_frame_nr / ... of one node instance lives on the GPU).
"""
import numpy as np

from . import person_msgs as pm


class PosePrior:
    def __init__(self, node, seq=0):
        self.node = node
        self.seq = int(seq)

    def skeleton_callback(self, persons, t, fb_delay_per_cam=None):
        return self.node.refine(self.seq, persons, t, fb_delay_per_cam)


def smoke_stage2(node, rig, frames, res, O):
    """Used by __graft_entry__.smoke(): refine the stage-1 persons of a short sequence and check against the oracle."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    from parity import compare_stage2
    node.refine_reset()
    pp = PosePrior(node, 0)
    ora = O.PosePrior(node.params)
    for fr, r in zip(frames, res):
        g = pp.skeleton_callback(r.published(), fr.stamp, fr.fb_delay)
        o = ora.step(r.published(), rig.n_cam, fr.stamp, fr.fb_delay, capacity=node.limits.max_hyp)
        compare_stage2(g, o)
    # feedback reprojection of the stage-1 persons of the last frame (pose_reprojection) against the oracle
    node.set_reprojection_cameras()
    got = node.reproject(res[-1].published())
    ref = O.reproject(rig, "simple", res[-1].published())
    assert np.array_equal(got.det_offset, ref.det_offset) and got.det_offset[-1] > 0
    n = int(ref.det_offset[-1])
    assert np.array_equal(got.person_index[:n], ref.person_index[:n])
    assert np.allclose(got.persons[:n]["keypoints"]["x"], ref.persons[:n]["keypoints"]["x"], rtol=2e-6, atol=1e-4)
