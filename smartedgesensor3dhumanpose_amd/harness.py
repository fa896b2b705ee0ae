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


class ApproximateTimeSync:
    """Python twin of host/se3d_sync.hpp (the reference's my_message_filters ApproximateTimeVec policy, SYNC:170-626): N FIFOs,
    candidate = FIFO heads, pivot = latest head of the first candidate, improvement test with the age penalty, optimality proof with
    the inter-message lower bound, overflow drop.  Messages are (stamp_ns, payload) pairs; `callback(list_of_messages)`."""

    def __init__(self, queue_size, num_inputs, callback, age_penalty=0.1, lower_bound_ns=0, max_interval_ns=None):
        self.qs, self.n, self.cb = queue_size, num_inputs, callback
        self.pen, self.max_interval = age_penalty, max_interval_ns
        self.fifo = [[] for _ in range(num_inputs)]
        self.past = [[] for _ in range(num_inputs)]
        self.dropped = [False] * num_inputs
        self.warned = [False] * num_inputs
        self.lb = [lower_bound_ns] * num_inputs
        self.non_empty, self.pivot, self.cand = 0, -1, None
        self.pivot_time = self.c_start = self.c_end = 0
        self.warnings = 0

    @staticmethod
    def _scaled(ns, scale):  # ros::Duration * double
        s, r = divmod(ns, 1000000000)
        d = (float(s) + 1e-9 * float(r)) * scale
        fl = np.floor(d)
        frac = (d - fl) * 1e9
        return int(fl) * 1000000000 + int(np.floor(frac + 0.5) if frac >= 0 else np.ceil(frac - 0.5))

    def _not_better(self, end, start_ref):
        return self._scaled(end - self.c_end, 1.0 + self.pen) >= start_ref - self.c_start

    def add(self, msg, i):
        q = self.fifo[i]
        q.append(msg)
        if len(q) == 1:
            self.non_empty += 1
            if self.non_empty == self.n:
                self._process()
        elif not self.warned[i]:
            now, prev = q[-1][0], q[-2][0]
            if now < prev or now - prev < self.lb[i]:
                self.warned[i] = True
                self.warnings += 1
        if len(q) + len(self.past[i]) > self.qs:
            self.non_empty = 0
            for j in range(self.n):
                self._restore(len(self.past[j]), j)
            q.pop(0)
            self.dropped[i] = True
            if self.pivot >= 0:
                self.cand, self.pivot = None, -1
                self._process()

    def _pop_front(self, i):
        self.fifo[i].pop(0)
        if not self.fifo[i]:
            self.non_empty -= 1

    def _front_to_past(self, i):
        self.past[i].append(self.fifo[i][0])
        self._pop_front(i)

    def _restore(self, count, i):
        for _ in range(count):
            self.fifo[i].insert(0, self.past[i].pop())
        if self.fifo[i]:
            self.non_empty += 1

    def _take(self):
        self.cand = [q[0] for q in self.fifo]
        for p in self.past:
            p.clear()

    def _publish(self):
        self.cb(self.cand)
        self.cand, self.pivot, self.non_empty = None, -1, 0
        for i in range(self.n):
            while self.past[i]:
                self.fifo[i].insert(0, self.past[i].pop())
            self.fifo[i].pop(0)
            if self.fifo[i]:
                self.non_empty += 1

    def _boundary(self, times, end):
        t, idx = times[0], 0
        for i, ti in enumerate(times):
            if i and ((ti < t) != end):
                t, idx = ti, i
        return idx, t

    def _virtual(self, i):
        if not self.fifo[i]:
            return max(self.past[i][-1][0] + self.lb[i], self.pivot_time)
        return self.fifo[i][0][0]

    def _process(self):
        while self.non_empty == self.n:
            heads = [q[0][0] for q in self.fifo]
            e_idx, e_t = self._boundary(heads, True)
            s_idx, s_t = self._boundary(heads, False)
            for i in range(self.n):
                if i != e_idx:
                    self.dropped[i] = False
            if self.pivot < 0:
                if (self.max_interval is not None and e_t - s_t > self.max_interval) or self.dropped[e_idx]:
                    self._pop_front(s_idx)
                    continue
                self._take()
                self.c_start, self.c_end, self.pivot, self.pivot_time = s_t, e_t, e_idx, e_t
                self._front_to_past(s_idx)
            elif self._not_better(e_t, s_t):
                self._front_to_past(s_idx)
            else:
                self._take()
                self.c_start, self.c_end = s_t, e_t
                self._front_to_past(s_idx)
            if s_idx == self.pivot:
                self._publish()
            elif self._not_better(e_t, self.pivot_time):
                self._publish()
            elif self.non_empty < self.n:
                moved = [0] * self.n
                while True:
                    vt = [self._virtual(i) for i in range(self.n)]
                    _, ve_t = self._boundary(vt, True)
                    vs_idx, vs_t = self._boundary(vt, False)
                    if self._not_better(ve_t, self.pivot_time):
                        self._publish()
                        break
                    if not self._not_better(ve_t, vs_t):
                        self.non_empty = 0
                        for i in range(self.n):
                            self._restore(moved[i], i)
                        break
                    self._front_to_past(vs_idx)
                    moved[vs_idx] += 1
