"""Host-side mirror of the reference's pose_prior node interface (pose_prior_mult_node.cpp) over the HIP C-ABI.

`PosePrior.skeleton_callback(persons, t)` corresponds to skeletonCallback(PersonCovList) (PP:505): it consumes the stage-1
persons of one frame and returns the persons3d_fused / persons3d_fused_pred payloads (PP:905-907).  One PosePrior = one
sequence slot of a Skeleton3D context (the global state g_tracks / g_frame_nr / ... of one node instance lives on the GPU).

The node's state can be taken out of the context and put into another one (`save_state` / `load_state`: se3d_refine_save / _load, a host blob
that does not depend on the context's limits — how a host layer moves a running sequence into a context with larger limits, the reference's
g_tracks growing without bound, PP:570-580), and read as a table (`tracks`: the TrackingHypothesis members of PP:68-76 per live track).
"""
import ctypes as C

import numpy as np

from . import person_msgs as pm

# layout of the blob (csrc/se3d_api.hip: SeqStateHeader, SeqStateTrack)
_header_dt = np.dtype([("magic", "<u4"), ("n_tracks", "<i4"), ("next_id", "<i4"), ("frame_nr", "<i4"), ("t_prev", "<f8"), ("fb_delay_buffer", "<f8", 3),
                       ("flags", "<u4"), ("_pad", "<u4")])
track_dt = np.dtype([("id", "<i4"), ("num_obs", "<i4"), ("mask", "<u4"), ("_pad", "<u4"), ("t_prev", "<f8"), ("height_prev", "<f8"), ("root_prev", "<f8", 3),
                     ("prev", "<f8", (pm.NUM_FUSION_KEYPOINTS, 3)), ("vel", "<f8", (pm.NUM_FUSION_KEYPOINTS, 3, 3))])
_MAGIC = 0x53325354  # "S2ST"


class PosePrior:
    def __init__(self, node, seq=0):
        self.node = node
        self.seq = int(seq)

    def skeleton_callback(self, persons, t, fb_delay_per_cam=None):
        return self.node.refine(self.seq, persons, t, fb_delay_per_cam)

    # ---- the node's global state (PP:54-60, 68-76, 123) ----
    def save_state(self):
        """The sequence's state as bytes (se3d_refine_save): header + one record per live track, in g_tracks order."""
        L, h = self.node._L, C.c_void_p(self.node._h)
        cap = int(L.se3d_refine_state_bytes(h))
        buf = (C.c_ubyte * cap)()
        got = C.c_int64(0)
        self.node._chk(L.se3d_refine_save(h, self.seq, buf, C.c_int64(cap), C.byref(got)), "se3d_refine_save")
        return bytes(buf)[: got.value]

    def load_state(self, blob):
        """Replace the sequence's state by a blob of save_state() — of this or of another context (any limits that hold its tracks)."""
        buf = (C.c_ubyte * len(blob)).from_buffer_copy(blob)
        self.node._chk(self.node._L.se3d_refine_load(C.c_void_p(self.node._h), self.seq, buf, C.c_int64(len(blob))), "se3d_refine_load")

    @staticmethod
    def parse_state(blob):
        """(header record, tracks [n] of track_dt): id, num_obs, mask of the joints in prevEstimate, t_prev, height_prev, root_prev,
        prev (the normalised previous estimate per fusion joint), vel (the three-frame velocity buffer per joint).  prev / vel are defined
        for the joints of `mask` only."""
        hd = np.frombuffer(blob, _header_dt, 1, 0)[0]
        if int(hd["magic"]) != _MAGIC:
            raise ValueError("not a pose_prior state blob")
        n = int(hd["n_tracks"])
        return hd, np.frombuffer(blob, track_dt, n, _header_dt.itemsize).copy()

    def tracks(self):
        """The live tracks of the sequence (g_tracks, PP:123) as a structured array of track_dt, list order."""
        return self.parse_state(self.save_state())[1]
