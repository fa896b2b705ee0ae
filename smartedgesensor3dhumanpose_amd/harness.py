"""ROS-free replay harness: the host bookkeeping of skeletonThreadCallback (skeleton_3d_triang_mult_node.cpp:1029-1084)
around a stage-1 backend, and the message plumbing of pose_prior's skeletonCallback (pose_prior_mult_node.cpp:528-532)
around a stage-2 backend.  Backends are callables, so the same harness drives the HIP nodes (product) and, in tests, the
CPU oracle.  host/se3d_nodes.cpp is the C++ twin of this file; host/se3d_replay reads the scene files written here.
"""
import struct
from dataclasses import dataclass, field

import numpy as np

from . import person_msgs as pm

G_MAX_SYNC_DIFF = 0.067  # S3D:64


@dataclass
class CameraMsg:
    """One Person2DList message of one camera."""
    stamp: float
    fb_delay: float = -1.0
    persons: np.ndarray = field(default_factory=lambda: np.zeros(0, pm.person2d_dt))


@dataclass
class Persons3D:
    """PersonCovList payload."""
    stamp: float
    ts_per_cam: list
    fb_delay_per_cam: list
    persons: np.ndarray


class Skeleton3DHarness:
    def __init__(self, n_cam, triangulate):
        """triangulate(frame: person_msgs.Frame) -> FrameResult-like with .published()"""
        self.n_cam = n_cam
        self.triangulate = triangulate
        self.last_stamp = 0.0
        self.log = []

    def step(self, msgs):
        """Returns Persons3D or None when the frame is skipped (delta_t <= 0, S3D:1043-1046)."""
        assert len(msgs) == self.n_cam
        t_max, idx = 0.0, -1
        for i, m in enumerate(msgs):  # pivot = most recent message (S3D:1029-1036)
            if m.stamp > t_max:
                t_max, idx = m.stamp, i
        if idx < 0:
            return None
        delta_t = msgs[idx].stamp - self.last_stamp
        if delta_t > 0.17:
            self.log.append("large frame delay %.3f" % delta_t)
        if delta_t <= 0.0:
            self.log.append("non-monotonic frame dropped")
            return None
        self.last_stamp = msgs[idx].stamp
        msgs = list(msgs)
        for i, m in enumerate(msgs):  # stale views are replaced by an empty dummy (S3D:1049-1057)
            if t_max - m.stamp > G_MAX_SYNC_DIFF:
                msgs[i] = CameraMsg(m.stamp, m.fb_delay, np.zeros(0, pm.person2d_dt))
                self.log.append("camera %d stale, removed" % i)
        off = np.zeros(self.n_cam + 1, np.int32)
        for i, m in enumerate(msgs):
            off[i + 1] = off[i] + len(m.persons)
        persons = np.concatenate([m.persons for m in msgs]) if off[-1] else np.zeros(0, pm.person2d_dt)
        frame = pm.Frame(self.n_cam, off, persons, np.array([m.fb_delay for m in msgs], np.float32), msgs[idx].stamp)
        res = self.triangulate(frame)
        return Persons3D(msgs[idx].stamp, [m.stamp for m in msgs], [float(np.float32(m.fb_delay)) for m in msgs], res.published().copy())


class PosePriorHarness:
    def __init__(self, refine):
        """refine(persons: person3d array, n_cam, t, fb_delay_per_cam) -> FusedResult-like"""
        self.refine = refine

    def callback(self, p3d: Persons3D):
        r = self.refine(p3d.persons, len(p3d.fb_delay_per_cam), p3d.stamp, np.array(p3d.fb_delay_per_cam, np.float32))
        n = r.n_persons
        fb = [float(r.c.pred_delta_t)] * len(p3d.fb_delay_per_cam)  # PP:531
        return (Persons3D(p3d.stamp, p3d.ts_per_cam, fb, r.fused[:n].copy()), Persons3D(p3d.stamp, p3d.ts_per_cam, fb, r.fused_pred[:n].copy()))


def stream_from_frames(frames):
    """Turns synchronised scene frames into per-camera message lists (all cameras stamped with the frame stamp)."""
    out = []
    for fr in frames:
        out.append([CameraMsg(fr.stamp, float(fr.fb_delay[c]), fr.persons[fr.det_offset[c]:fr.det_offset[c + 1]].copy()) for c in range(fr.n_cam)])
    return out


# ---- scene / result files of host/se3d_replay ---------------------------------------------------------------------
def write_replay_scene(path, rig, params, stream):
    with open(path, "wb") as f:
        f.write(b"SE3DSCN1")
        f.write(struct.pack("<i", rig.n_cam))
        for c in range(rig.n_cam):
            f.write(np.ascontiguousarray(rig.T[c], "<f8").tobytes())
            f.write(np.asarray(rig.K[c], "<f8").tobytes())
        f.write(struct.pack("<iid", params.pose_method, params.norm_height, params.max_epi_dist))
        f.write(struct.pack("<i", len(stream)))
        for msgs in stream:
            for m in msgs:
                f.write(struct.pack("<dfi", m.stamp, m.fb_delay, len(m.persons)))
                for p in m.persons:
                    f.write(struct.pack("<f", p["score"]))
                    f.write(np.asarray(p["bbox"], "<f4").tobytes())
                    kp = p["keypoints"]
                    rec = np.zeros((pm.NUM_KEYPOINTS, 6), "<f4")
                    rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3:] = kp["x"], kp["y"], kp["score"], kp["cov"]
                    f.write(rec.tobytes())


def _read_list(buf, pos):
    stamp, nfb = struct.unpack_from("<di", buf, pos)
    pos += 12
    fb = list(struct.unpack_from("<%df" % nfb, buf, pos))
    pos += 4 * nfb
    ts = list(struct.unpack_from("<%dd" % nfb, buf, pos))
    pos += 8 * nfb
    (n,) = struct.unpack_from("<i", buf, pos)
    pos += 4
    persons = np.zeros(n, pm.person3d_dt)
    for i in range(n):
        (persons[i]["id"],) = struct.unpack_from("<I", buf, pos)
        pos += 4
        for k in range(pm.NUM_FUSION_KEYPOINTS):
            v = struct.unpack_from("<3df6d", buf, pos)
            pos += 76
            persons[i]["keypoints"][k]["joint"] = v[:3]
            persons[i]["keypoints"][k]["score"] = v[3]
            persons[i]["keypoints"][k]["cov"] = v[4:]
    return Persons3D(stamp, ts, fb, persons), pos


def read_replay_output(path):
    buf = open(path, "rb").read()
    (n_frames,) = struct.unpack_from("<i", buf, 0)
    pos, out = 4, []
    for _ in range(n_frames):
        (processed,) = struct.unpack_from("<i", buf, pos)
        pos += 4
        if not processed:
            out.append(None)
            continue
        p3d, pos = _read_list(buf, pos)
        fused, pos = _read_list(buf, pos)
        pred, pos = _read_list(buf, pos)
        out.append((p3d, fused, pred))
    return out


def read_replay_reproj(path):
    """Per-frame, per-camera (stamp, fb_delay, persons[person2d_dt]) written by `se3d_replay scene out reproj.bin`; None = not published."""
    buf = open(path, "rb").read()
    n_frames, n_cam = struct.unpack_from("<ii", buf, 0)
    pos, out = 8, []
    while pos < len(buf):
        (ok,) = struct.unpack_from("<i", buf, pos)
        pos += 4
        if not ok:
            out.append(None)
            continue
        cams = []
        for _ in range(n_cam):
            stamp, fb, n = struct.unpack_from("<dfi", buf, pos)
            pos += 16
            persons = np.frombuffer(buf, pm.person2d_dt, n, pos).copy()
            pos += n * pm.person2d_dt.itemsize
            cams.append((stamp, fb, persons))
        out.append(cams)
    return out
