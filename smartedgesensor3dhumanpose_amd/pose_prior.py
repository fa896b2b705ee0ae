"""Host-side mirror of the reference's pose_prior node interface (pose_prior_mult_node.cpp) over the HIP C-ABI.

`PosePrior.skeleton_callback(persons, t)` corresponds to skeletonCallback(PersonCovList) (PP:505): it consumes the stage-1
persons of one frame and returns the persons3d_fused / persons3d_fused_pred payloads (PP:905-907).  One PosePrior = one
sequence slot of a Skeleton3D context (the global state g_tracks / g_frame_nr / ... of one node instance lives on the GPU).
"""

class PosePrior:
    def __init__(self, node, seq=0):
        self.node = node
        self.seq = int(seq)

    def skeleton_callback(self, persons, t, fb_delay_per_cam=None):
        return self.node.refine(self.seq, persons, t, fb_delay_per_cam)
