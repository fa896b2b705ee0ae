"""Flat POD mirrors of the reference's person_msgs/*.msg and of include/se3d.h (numpy dtypes + ctypes structs).

person_msgs/msg/Keypoint2D.msg, Person2D.msg, Person2DList.msg, KeypointWithCovariance.msg, PersonCov.msg,
PersonCovList.msg are the public API surface of every reference node.  The C-ABI uses fixed-size
equivalents; these definitions must stay byte-identical with include/se3d.h (checked in tests/test_abi.py).
"""
import ctypes as C

import numpy as np

NUM_KEYPOINTS = 17
NUM_FUSION_KEYPOINTS = 21
POSE_SIMPLE, POSE_H36M = 0, 1
S2FLAG_TRACK_OVERFLOW, S2FLAG_ASSIGNMENT_FAILED, S2FLAG_NOT_A_FOREST = 1, 4, 8   # se3d_fused_out.flags (sticky per sequence)

FLAG_HYP_OVERFLOW, FLAG_FEW_CAMERAS, FLAG_ASSIGNMENT_FAILED = 1, 2, 4

keypoint2d_dt = np.dtype([("x", "<f4"), ("y", "<f4"), ("score", "<f4"), ("cov", "<f4", (3,))])
person2d_dt = np.dtype([("score", "<f4"), ("bbox", "<f4", (4,)), ("keypoints", keypoint2d_dt, (NUM_KEYPOINTS,))])
keypoint3d_dt = np.dtype([("joint", "<f8", (3,)), ("score", "<f4"), ("_pad", "<f4"), ("cov", "<f8", (6,))])
person3d_dt = np.dtype([("id", "<u4"), ("score", "<f4"), ("keypoints", keypoint3d_dt, (NUM_FUSION_KEYPOINTS,))])

assert keypoint2d_dt.itemsize == 24 and person2d_dt.itemsize == 428
assert keypoint3d_dt.itemsize == 80 and person3d_dt.itemsize == 1688

# FUSION_BODY_PARTS (skeleton_3d/include/skeleton_3d/fusion_body_parts.h:3-24)
FUSION = dict(Nose=0, Neck=1, RShoulder=2, RElbow=3, RWrist=4, LShoulder=5, LElbow=6, LWrist=7, MidHip=8, RHip=9,
              RKnee=10, RAnkle=11, LHip=12, LKnee=13, LAnkle=14, REye=15, LEye=16, REar=17, LEar=18, Head=19, Belly=20)
# g_kp2kpFusion_idx_simple (skeleton_3d_triang_mult_node.cpp:139-142)
KP2FUSION_SIMPLE = [0, 16, 15, 18, 17, 5, 2, 6, 3, 7, 4, 12, 9, 13, 10, 14, 11]


class Rig(C.Structure):
    _fields_ = [("n_cam", C.c_int32), ("Rt", C.POINTER(C.c_float)), ("K", C.POINTER(C.c_float)), ("F", C.POINTER(C.c_float))]


class Params(C.Structure):
    _fields_ = [("pose_method", C.c_int32), ("norm_height", C.c_int32), ("max_epi_dist", C.c_double)]


ABI_VERSION = 6   # SE3D_ABI_VERSION of include/se3d.h the ctypes mirrors below follow (checked against the loaded library by _lib.lib())


class Limits(C.Structure):
    _fields_ = [("max_frames", C.c_int32), ("max_det_per_cam", C.c_int32), ("max_hyp", C.c_int32), ("max_tracks", C.c_int32)]


class Options(C.Structure):
    """se3d_options (include/se3d.h): all zero = the library's defaults.  Symbolic values: the tables below."""
    _fields_ = [(n, C.c_int32) for n in ("pair_chunks", "hi_priority_stream", "reserve_cus", "graph", "walk_shape", "sweep_lds_kb",
                                         "pairs_form", "sweep_form", "tri_form", "post_form", "lm_form", "walk_form", "walk_lm", "walk_book",
                                         "refine_steps", "munkres_form", "reproj_form", "stream_out", "exact_cov")] + [("reserved", C.c_int32 * 5)]


OPTION_VALUES = {   # names of the SE3D_* constants of include/se3d.h
    "walk_shape": {"auto": 0, "wave": 1, "wg": 2},
    "pairs_form": {"auto": 0, "flat": 1, "rows": 2, "fused": 3},
    "sweep_form": {"auto": 0, "per_camera": 1, "wave_global": 2},
    "tri_form": {"auto": 0, "joint": 1},
    "post_form": {"auto": 0, "block": 1},
    "lm_form": {"tree": 0, "tree_mfma": 1, "dense": 2},
    "walk_form": {"fused": 0, "steps": 1},
    "walk_lm": {"trio": 0, "solo": 1},
    "munkres_form": {"auto": 0, "block": 1, "wave": 2},
    "exact_cov": {"auto": 0, "f32": 1},
}


def make_options(base=None, **kw):
    """Options from keyword arguments; values may be the integers of include/se3d.h or the names in OPTION_VALUES
    (e.g. lm_form="dense", hi_priority_stream=-1)."""
    o = Options()
    if base is not None:
        C.memmove(C.byref(o), C.byref(base), C.sizeof(Options))
    names = {n for n, _ in Options._fields_} - {"reserved"}
    for k, v in kw.items():
        if k not in names:
            raise KeyError("unknown se3d option %r" % k)
        if isinstance(v, str):
            v = OPTION_VALUES[k][v]
        setattr(o, k, int(v))
    return o


class FrameIn(C.Structure):
    _fields_ = [("n_cam", C.c_int32), ("det_offset", C.POINTER(C.c_int32)), ("persons", C.c_void_p),
                ("fb_delay", C.POINTER(C.c_float)), ("stamp", C.c_double)]


class FrameOut(C.Structure):
    _fields_ = [("capacity", C.c_int32), ("hyp_capacity", C.c_int32), ("n_persons", C.c_int32), ("n_hyp", C.c_int32),
                ("flags", C.c_uint32), ("persons", C.c_void_p), ("person_hyp", C.POINTER(C.c_int32)),
                ("hyp_assoc", C.POINTER(C.c_int32))]


class FusedOut(C.Structure):
    _fields_ = [("capacity", C.c_int32), ("n_persons", C.c_int32), ("n_tracks", C.c_int32), ("pred_delta_t", C.c_float),
                ("fused", C.c_void_p), ("fused_pred", C.c_void_p), ("lm_iterations", C.POINTER(C.c_int32)), ("flags", C.c_uint32),
                ("lm_objective", C.POINTER(C.c_double))]


class ReprojOut(C.Structure):
    _fields_ = [("capacity", C.c_int32), ("det_offset", C.POINTER(C.c_int32)), ("persons", C.c_void_p), ("person_index", C.POINTER(C.c_int32))]


class StreamResult(C.Structure):
    _fields_ = [("n_frames", C.c_int32), ("n_persons", C.c_int32), ("person_offset", C.POINTER(C.c_int32)), ("fused", C.c_void_p), ("fused_pred", C.c_void_p)]


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class Frame:
    """One synchronised set of per-camera Person2DList payloads (ragged), kept alive for the C call."""

    def __init__(self, n_cam, det_offset, persons, fb_delay=None, stamp=0.0):
        self.n_cam = int(n_cam)
        self.det_offset = np.ascontiguousarray(det_offset, dtype=np.int32)
        self.persons = np.ascontiguousarray(persons, dtype=person2d_dt)
        self.fb_delay = (np.full(n_cam, -1.0, np.float32) if fb_delay is None else np.ascontiguousarray(fb_delay, np.float32))
        self.stamp = float(stamp)
        assert self.det_offset.shape == (n_cam + 1,) and self.det_offset[-1] == len(self.persons)

    def as_c(self):
        return FrameIn(self.n_cam, iptr(self.det_offset), self.persons.ctypes.data, fptr(self.fb_delay), self.stamp)

    def n_det(self, cam):
        return int(self.det_offset[cam + 1] - self.det_offset[cam])


class FrameResult:
    """Caller-owned output buffers of one stage-1 call."""

    def __init__(self, n_cam, capacity, hyp_capacity):
        self.n_cam = n_cam
        self.persons = np.zeros(capacity, person3d_dt)
        self.person_hyp = np.full(capacity, -1, np.int32)
        self.hyp_assoc = np.full((max(hyp_capacity, 1), n_cam), -1, np.int32)
        self.c = FrameOut(capacity, hyp_capacity, 0, 0, 0, self.persons.ctypes.data, iptr(self.person_hyp), iptr(self.hyp_assoc))

    @property
    def n_persons(self):
        return self.c.n_persons

    @property
    def n_hyp(self):
        return self.c.n_hyp

    @property
    def flags(self):
        return self.c.flags

    def published(self):
        return self.persons[: self.c.n_persons]

    def assoc(self):
        return self.hyp_assoc[: min(self.c.n_hyp, self.c.hyp_capacity)]


class FusedResult:
    """Caller-owned output buffers of one stage-2 call."""

    def __init__(self, capacity):
        self.fused = np.zeros(capacity, person3d_dt)
        self.fused_pred = np.zeros(capacity, person3d_dt)
        self.lm_iterations = np.zeros(capacity, np.int32)
        self.lm_objective = np.zeros(capacity, np.float64)   # graph.error(result) per published person (se3d_fused_out.lm_objective; -1: optimisation failed)
        self.c = FusedOut(capacity, 0, 0, 0.0, self.fused.ctypes.data, self.fused_pred.ctypes.data, iptr(self.lm_iterations), 0,
                          self.lm_objective.ctypes.data_as(C.POINTER(C.c_double)))

    @property
    def n_persons(self):
        return self.c.n_persons


class ReprojResult:
    """Per-camera Person2DList payloads of the reprojection node (skeleton_reproj_mult_node.cpp:226-233)."""

    def __init__(self, n_cam, capacity):
        self.n_cam = n_cam
        self.det_offset = np.zeros(n_cam + 1, np.int32)
        self.persons = np.zeros(max(capacity, 1), person2d_dt)
        self.person_index = np.full(max(capacity, 1), -1, np.int32)
        self.c = ReprojOut(int(capacity), iptr(self.det_offset), self.persons.ctypes.data, iptr(self.person_index))

    def camera(self, i):
        a, b = int(self.det_offset[i]), int(self.det_offset[i + 1])
        return self.persons[a:b], self.person_index[a:b]
