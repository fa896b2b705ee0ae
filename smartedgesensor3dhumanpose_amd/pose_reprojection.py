"""Host-side mirror of the reference's pose_reprojection node (skeleton_reproj_mult_node.cpp) over the HIP C-ABI.

`SkeletonReprojection.fused_skeleton_callback(persons, ts_per_cam, fb_delay_per_cam)` corresponds to
fusedSkeletonCallback(PersonCovList) (REPROJ:139): it projects the fused / predicted 3-D skeletons with their covariances into
every camera and returns one Person2DList payload per camera (published on <cam>/skel_pred, REPROJ:234).
"""
import numpy as np

from . import person_msgs as pm


class Person2DList:
    """person_msgs/Person2DList.msg: header.stamp, fb_delay, persons[]."""

    def __init__(self, stamp, fb_delay, persons, person_index):
        self.stamp, self.fb_delay, self.persons, self.person_index = stamp, fb_delay, persons, person_index


class SkeletonReprojection:
    def __init__(self, node, T_base_to_cam=None, P=None, wh=None):
        self.node = node
        node.set_reprojection_cameras(T_base_to_cam, P, wh)

    def fused_skeleton_callback(self, persons, ts_per_cam=None, fb_delay_per_cam=None, frame_id="base"):
        n = self.node.rig.n_cam
        if frame_id != "base":  # REPROJ:141-143
            raise ValueError("fused_skeleton_callback: frame_id must be 'base', got %r" % (frame_id,))
        if ts_per_cam is not None and len(ts_per_cam) != n:  # REPROJ:147
            raise ValueError("ts_per_cam must have one entry per camera")
        res = self.node.reproject(np.ascontiguousarray(persons, pm.person3d_dt))
        out = []
        for i in range(n):
            p, idx = res.camera(i)
            out.append(Person2DList(None if ts_per_cam is None else ts_per_cam[i], None if fb_delay_per_cam is None else fb_delay_per_cam[i],
                                    p.copy(), idx.copy()))
        return out
