"""Host-side mirror of the reference's skeleton_3d node interface over the HIP C-ABI.

Names and argument meaning follow skeleton_3d_triang_mult_node.cpp: ROS private params `pose_method`, `max_epi_dist`
(S3D:1095-1097), entry `triangulate_persons(people, ...)` (S3D:525).  All arithmetic happens in libse3d_hip.so on the GPU.
"""
import ctypes as C

import numpy as np

from . import _lib
from . import person_msgs as pm


class Se3dError(RuntimeError):
    pass


def rig_tensors(rig):
    """Product-side host set-up of Rt / F (se3d_rig_from_transforms, replaces S3D:1187-1211)."""
    Rt = np.zeros((rig.n_cam, 12), np.float32)
    F = np.zeros((rig.n_pairs, 9), np.float32)
    rc = _lib.lib().se3d_rig_from_transforms(rig.n_cam, rig.T.ctypes.data_as(C.POINTER(C.c_double)), pm.fptr(Rt), pm.fptr(F))
    if rc != 0:
        raise Se3dError("se3d_rig_from_transforms failed: %d" % rc)
    return Rt, F


def device_bytes(n_cam, max_frames, max_det_per_cam, max_hyp, max_tracks=None, options=None):
    """Device memory a context of these limits (and options: a dict or pm.Options) allocates at creation (se3d_device_bytes_ex: host
    arithmetic, no GPU call)."""
    lim = pm.Limits(int(max_frames), int(max_det_per_cam), int(max_hyp), int(max_tracks or max_hyp))
    opt = options if isinstance(options, pm.Options) else pm.make_options(**(options or {}))
    n = _lib.lib().se3d_device_bytes_ex(int(n_cam), C.byref(lim), C.byref(opt))
    if n < 0:
        raise Se3dError("se3d_device_bytes: unsupported sizes")
    return int(n)


class Skeleton3D:
    """One skeleton_3d node instance bound to one GPU."""

    def __init__(self, rig, pose_method="simple", max_epi_dist=0.050, norm_height=False, max_frames=1, max_det_per_cam=8, max_hyp=None,
                 max_tracks=None, device=0, options=None):
        """options: dict of se3d_options fields (include/se3d.h; values as integers or the names of person_msgs.OPTION_VALUES), e.g.
        dict(hi_priority_stream=-1) for more than four large-rig contexts per GPU; None = the library's defaults."""
        if rig.Rt is None or rig.F is None:
            rig.set_tensors(*rig_tensors(rig))
        self.rig = rig
        self.params = pm.Params(pm.POSE_H36M if pose_method == "h36m" else pm.POSE_SIMPLE, int(bool(norm_height)), float(max_epi_dist))
        max_hyp = max_hyp or min(256, max(2 * max_det_per_cam, 8))
        self.limits = pm.Limits(int(max_frames), int(max_det_per_cam), int(max_hyp), int(max_tracks or max_hyp))
        self._rig_c = pm.Rig(rig.n_cam, pm.fptr(rig.Rt), pm.fptr(rig.K), pm.fptr(rig.F))
        self._L = _lib.lib()
        self.options = options if isinstance(options, pm.Options) else pm.make_options(**(options or {}))
        self._h = self._L.se3d_create_ex(C.byref(self._rig_c), C.byref(self.params), C.byref(self.limits), C.byref(self.options), int(device))
        if not self._h:
            why = self._L.se3d_last_error(None)
            raise Se3dError("se3d_create failed: %s; there is no CPU fallback" % (why.decode() if why else "no GPU, unsupported sizes or out of memory"))
        self._staged = 0
        self.device = int(device)

    def set_options(self, **kw):
        """Change launch-time options of the live context (se3d_set_options), e.g. set_options(lm_form="dense"); pair_chunks,
        hi_priority_stream and reserve_cus are fixed at creation."""
        opt = pm.make_options(self.options, **kw)
        self._chk(self._L.se3d_set_options(C.c_void_p(self._h), C.byref(opt)), "se3d_set_options")
        self.options = opt

    # -- error helper
    def _chk(self, rc, what):
        if rc != 0:
            raise Se3dError("%s failed (%d): %s" % (what, rc, self._L.se3d_last_error(C.c_void_p(self._h)).decode()))

    # -- reference entry point: triangulate_persons for a batch of synchronised frames
    def triangulate_persons(self, frames, capacity=None):
        frames = list(frames)
        arr = (pm.FrameIn * len(frames))(*[f.as_c() for f in frames])
        res = [pm.FrameResult(self.rig.n_cam, capacity or self.limits.max_hyp, self.limits.max_hyp) for _ in frames]
        outs = (pm.FrameOut * len(frames))(*[r.c for r in res])
        self._chk(self._L.se3d_triangulate_batch(C.c_void_p(self._h), len(frames), arr, outs), "se3d_triangulate_batch")
        for r, o in zip(res, outs):
            r.c = o
        return res

    # -- config 2: batched DLT triangulation with the association given (rows N, T, P, O, U)
    def triangulate_assoc(self, frames, assoc):
        """assoc: [n_frames][n_hyp][n_cam] original detection indices or -1 -> person3d array [n_frames][n_hyp]."""
        frames = list(frames)
        assoc = np.ascontiguousarray(assoc, np.int32).reshape(len(frames), -1, self.rig.n_cam)
        arr = (pm.FrameIn * len(frames))(*[f.as_c() for f in frames])
        out = np.zeros((len(frames), assoc.shape[1]), pm.person3d_dt)
        self._chk(self._L.se3d_triangulate_assoc(C.c_void_p(self._h), len(frames), arr, assoc.shape[1], pm.iptr(assoc), C.c_void_p(out.ctypes.data)),
                  "se3d_triangulate_assoc")
        return out

    def stage_assoc(self, assoc):
        assoc = np.ascontiguousarray(assoc, np.int32).reshape(self._staged, -1, self.rig.n_cam)
        self._dlt_hyp = assoc.shape[1]
        self._chk(self._L.se3d_stage_assoc(C.c_void_p(self._h), assoc.shape[1], pm.iptr(assoc)), "se3d_stage_assoc")

    def run_dlt(self):
        self._chk(self._L.se3d_run_dlt(C.c_void_p(self._h)), "se3d_run_dlt")

    def fetch_dlt(self, n_frames=None):
        n = n_frames or self._staged
        out = np.zeros((n, self._dlt_hyp), pm.person3d_dt)
        self._chk(self._L.se3d_fetch_dlt(C.c_void_p(self._h), n, C.c_void_p(out.ctypes.data)), "se3d_fetch_dlt")
        return out

    # -- hypothesis shard of one stream over the ranks: replicated association, hypotheses / detections split by index % world
    def set_shard(self, rank, world):
        self._chk(self._L.se3d_set_shard(C.c_void_p(self._h), int(rank), int(world)), "se3d_set_shard")

    def shard_bytes(self, stage, n):
        return int(self._L.se3d_shard_bytes(C.c_void_p(self._h), int(stage), int(n)))

    def run_stage1_shard_pre(self, pack_ptr):
        self._chk(self._L.se3d_run_stage1_shard_pre(C.c_void_p(self._h), C.c_void_p(pack_ptr)), "se3d_run_stage1_shard_pre")

    def run_stage1_shard_post(self, gathered_ptr):
        self._chk(self._L.se3d_run_stage1_shard_post(C.c_void_p(self._h), C.c_void_p(gathered_ptr)), "se3d_run_stage1_shard_post")

    def run_stage2_shard_pre(self, t, frame_slot0, pack_ptr):
        t = np.ascontiguousarray(t, np.float64)
        self._chk(self._L.se3d_run_stage2_shard_pre(C.c_void_p(self._h), len(t), int(frame_slot0), t.ctypes.data_as(C.POINTER(C.c_double)), C.c_void_p(pack_ptr)),
                  "se3d_run_stage2_shard_pre")

    def run_stage2_shard_post(self, n_seq, gathered_ptr):
        self._chk(self._L.se3d_run_stage2_shard_post(C.c_void_p(self._h), int(n_seq), C.c_void_p(gathered_ptr)), "se3d_run_stage2_shard_post")

    # -- streaming ingest: a batch of frames from the host every step, two batches in flight
    def stream_prepare(self, frames):
        """ctypes array of the frames (kept alive by the caller): building it is host bookkeeping of the caller, not part of the path."""
        frames = list(frames)
        return (pm.FrameIn * len(frames))(*[f.as_c() for f in frames]), frames

    def stream_submit(self, prepared, t, frame_slot0):
        arr, frames = prepared
        t = np.ascontiguousarray(t, np.float64)
        slots = np.ascontiguousarray(frame_slot0, np.int32)
        assert t.ndim == 2 and len(slots) == t.shape[0]
        self._chk(self._L.se3d_stream_submit(C.c_void_p(self._h), len(frames), arr, t.shape[1], t.shape[0], pm.iptr(slots),
                                             t.ctypes.data_as(C.POINTER(C.c_double))), "se3d_stream_submit")

    def stream_collect(self, copy=True):
        """Oldest batch in flight -> (person_offset [n_frames + 1], fused [n], fused_pred [n]); views into the context's pinned memory
        unless copy."""
        r = pm.StreamResult()
        self._chk(self._L.se3d_stream_collect(C.c_void_p(self._h), C.byref(r)), "se3d_stream_collect")
        off = np.ctypeslib.as_array(r.person_offset, shape=(r.n_frames + 1,))
        n = int(r.n_persons)
        if n == 0:
            return off.copy(), np.zeros(0, pm.person3d_dt), np.zeros(0, pm.person3d_dt)
        fused = np.frombuffer((C.c_char * (n * pm.person3d_dt.itemsize)).from_address(r.fused), dtype=pm.person3d_dt)
        pred = np.frombuffer((C.c_char * (n * pm.person3d_dt.itemsize)).from_address(r.fused_pred), dtype=pm.person3d_dt)
        return (off.copy(), fused.copy(), pred.copy()) if copy else (off, fused, pred)

    def copy_stream_probe(self):
        """Diagnostic: do the device's shared upload and download streams overlap?  dict(both_GBps: ~48 healthy / ~28 serialised, repairs: how often
        the library replaced the pair in this process, last_download_GBps)."""
        r, last = C.c_double(0.0), C.c_double(0.0)
        n = C.c_int32(0)
        self._chk(self._L.se3d_copy_stream_probe(C.c_int32(self.device), C.byref(r), C.byref(n), C.byref(last)), "se3d_copy_stream_probe")
        return {"both_GBps": round(r.value, 1), "repairs": int(n.value), "last_download_GBps": round(last.value, 1)}

    # -- resident pipeline
    def stage(self, frames):
        frames = list(frames)
        arr = (pm.FrameIn * len(frames))(*[f.as_c() for f in frames])
        self._chk(self._L.se3d_stage_frames(C.c_void_p(self._h), len(frames), arr), "se3d_stage_frames")
        self._staged = len(frames)

    def run_stage1(self):
        self._chk(self._L.se3d_run_stage1(C.c_void_p(self._h)), "se3d_run_stage1")

    def run_stage1_phases(self, phases):
        """1 normalise + pair table, 8 camera sweep, 2 triangulation, 4 post-processing (bit mask, executed in that order)."""
        self._chk(self._L.se3d_run_stage1_phases(C.c_void_p(self._h), int(phases)), "se3d_run_stage1_phases")

    def stream_handle(self):
        """hipStream_t of this context as an integer (e.g. for torch.cuda.ExternalStream)."""
        return int(self._L.se3d_stream_handle(C.c_void_p(self._h)) or 0)

    def sync(self):
        self._chk(self._L.se3d_sync(C.c_void_p(self._h)), "se3d_sync")

    def fetch_stage1(self, n_frames=None, capacity=None):
        n = n_frames or self._staged
        res = [pm.FrameResult(self.rig.n_cam, capacity or self.limits.max_hyp, self.limits.max_hyp) for _ in range(n)]
        outs = (pm.FrameOut * n)(*[r.c for r in res])
        self._chk(self._L.se3d_fetch_stage1(C.c_void_p(self._h), n, outs), "se3d_fetch_stage1")
        for r, o in zip(res, outs):
            r.c = o
        return res

    # -- stage 2 (pose_prior) on the same context: sequence s <-> frame slot s
    def refine_reset(self):
        self._chk(self._L.se3d_refine_reset(C.c_void_p(self._h)), "se3d_refine_reset")

    def refine(self, seq, dets, t, fb_delay_per_cam=None, capacity=None):
        """One pose_prior skeletonCallback (PP:505) for sequence `seq` with host measurements."""
        dets = np.ascontiguousarray(dets, pm.person3d_dt)
        res = pm.FusedResult(capacity or self.limits.max_hyp)
        fb = None if fb_delay_per_cam is None else np.ascontiguousarray(fb_delay_per_cam, np.float32)
        self._chk(self._L.se3d_refine(C.c_void_p(self._h), int(seq), len(dets), C.c_void_p(dets.ctypes.data), self.rig.n_cam,
                                      None if fb is None else pm.fptr(fb), C.c_double(t), C.byref(res.c)), "se3d_refine")
        return res

    def run_stage2(self, t, frame_slot0=0):
        t = np.ascontiguousarray(t, np.float64)
        self._chk(self._L.se3d_run_stage2(C.c_void_p(self._h), len(t), int(frame_slot0), t.ctypes.data_as(C.POINTER(C.c_double))), "se3d_run_stage2")

    def fetch_stage2(self, n_seq, capacity=None):
        res = [pm.FusedResult(capacity or self.limits.max_hyp) for _ in range(n_seq)]
        outs = (pm.FusedOut * n_seq)(*[r.c for r in res])
        self._chk(self._L.se3d_fetch_stage2(C.c_void_p(self._h), n_seq, outs), "se3d_fetch_stage2")
        for r, o in zip(res, outs):
            r.c = o
        return res

    def run_stage2_walk(self, t, frame_slot0):
        """t: [n_steps][n_seq] stamps; frame_slot0: [n_steps] first frame slot of each step.  One launch for the whole walk."""
        t = np.ascontiguousarray(t, np.float64)
        slots = np.ascontiguousarray(frame_slot0, np.int32)
        assert t.ndim == 2 and len(slots) == t.shape[0]
        self._chk(self._L.se3d_run_stage2_walk(C.c_void_p(self._h), t.shape[1], t.shape[0], pm.iptr(slots), t.ctypes.data_as(C.POINTER(C.c_double))),
                  "se3d_run_stage2_walk")

    def fetch_stage2_walk(self, frame_slot0, n_seq, capacity=None):
        res = [pm.FusedResult(capacity or self.limits.max_hyp) for _ in range(n_seq)]
        outs = (pm.FusedOut * n_seq)(*[r.c for r in res])
        self._chk(self._L.se3d_fetch_stage2_walk(C.c_void_p(self._h), int(frame_slot0), n_seq, outs), "se3d_fetch_stage2_walk")
        for r, o in zip(res, outs):
            r.c = o
        return res

    def pack_stage2_walk_device(self, n_frames, persons_ptr, counts_ptr):
        self._chk(self._L.se3d_pack_stage2_walk_device(C.c_void_p(self._h), int(n_frames), C.c_void_p(persons_ptr), C.c_void_p(counts_ptr)),
                  "se3d_pack_stage2_walk_device")

    def pack_stage2_walk_device_slots(self, n_frames, slots, persons_ptr, counts_ptr):
        """[n_frames][slots] persons (slots <= max_hyp) + [n_frames] true counts; asynchronous."""
        self._chk(self._L.se3d_pack_stage2_walk_device_slots(C.c_void_p(self._h), int(n_frames), int(slots), C.c_void_p(persons_ptr), C.c_void_p(counts_ptr)),
                  "se3d_pack_stage2_walk_device_slots")

    def pack_stage2_walk_device_dense(self, n_frames, capacity, persons_ptr, offsets_ptr):
        """Published fused persons back to back ([capacity] persons) + [n_frames + 2] int32 offsets (…, written, dropped); asynchronous."""
        self._chk(self._L.se3d_pack_stage2_walk_device_dense(C.c_void_p(self._h), int(n_frames), int(capacity), C.c_void_p(persons_ptr), C.c_void_p(offsets_ptr)),
                  "se3d_pack_stage2_walk_device_dense")

    def pack_stage2_device(self, n_seq, persons_ptr, counts_ptr):
        """Asynchronous D2D copy of the last run_stage2 results ([n_seq][max_hyp] persons, [n_seq] counts); sync() before use."""
        self._chk(self._L.se3d_pack_stage2_device(C.c_void_p(self._h), int(n_seq), C.c_void_p(persons_ptr), C.c_void_p(counts_ptr)), "se3d_pack_stage2_device")

    def pack_stage1_device(self, persons_ptr, counts_ptr):
        self._chk(self._L.se3d_pack_stage1_device(C.c_void_p(self._h), C.c_void_p(persons_ptr), C.c_void_p(counts_ptr)), "se3d_pack_stage1_device")

    # -- measurement
    # -- feedback reprojection (pose_reprojection node)
    def set_reprojection_cameras(self, T=None, P=None, wh=None):
        rig = self.rig
        T = np.ascontiguousarray(rig.T if T is None else T, np.float64).reshape(rig.n_cam, 16)
        P = np.ascontiguousarray(rig.K if P is None else P, np.float64).reshape(rig.n_cam, 4)
        if wh is None:
            wh = np.broadcast_to(np.array([rig.width, rig.height]), (rig.n_cam, 2))
        wh = np.ascontiguousarray(wh, np.int32).reshape(rig.n_cam, 2)
        self._chk(self._L.se3d_set_reprojection_cameras(C.c_void_p(self._h), T.ctypes.data_as(C.POINTER(C.c_double)),
                                                        P.ctypes.data_as(C.POINTER(C.c_double)), pm.iptr(wh)), "se3d_set_reprojection_cameras")

    def reproject(self, persons):
        persons = np.ascontiguousarray(persons, pm.person3d_dt)
        res = pm.ReprojResult(self.rig.n_cam, self.rig.n_cam * max(len(persons), 1))
        self._chk(self._L.se3d_reproject(C.c_void_p(self._h), len(persons), C.c_void_p(persons.ctypes.data), C.byref(res.c)), "se3d_reproject")
        return res

    def reproject_resident(self, seq=0, fetch=True):
        if not fetch:
            self._chk(self._L.se3d_reproject_resident(C.c_void_p(self._h), int(seq), None), "se3d_reproject_resident")
            return None
        res = pm.ReprojResult(self.rig.n_cam, self.rig.n_cam * self.limits.max_hyp)
        self._chk(self._L.se3d_reproject_resident(C.c_void_p(self._h), int(seq), C.byref(res.c)), "se3d_reproject_resident")
        return res

    def reproject_resident_batch(self, slot0, n_slots):
        """All result slots of a resident walk in one launch (asynchronous); fetch_reprojection(slot) returns one slot's lists."""
        self._chk(self._L.se3d_reproject_resident_batch(C.c_void_p(self._h), int(slot0), int(n_slots)), "se3d_reproject_resident_batch")

    def fetch_reprojection(self, slot):
        res = pm.ReprojResult(self.rig.n_cam, self.rig.n_cam * self.limits.max_hyp)
        self._chk(self._L.se3d_fetch_reprojection(C.c_void_p(self._h), int(slot), C.byref(res.c)), "se3d_fetch_reprojection")
        return res

    def fusion_constants(self):
        """(indices22, vel_sigmas21) compiled into the kernels (diagnostic; compared with the reference's fusion_body_parts.h by the tests)."""
        idx = (C.c_int32 * 22)()
        vel = (C.c_double * 21)()
        self._chk(self._L.se3d_get_fusion_constants(C.c_void_p(self._h), idx, vel), "se3d_get_fusion_constants")
        return list(idx), list(vel)

    def selftest_exact_math(self, n_div=1 << 33):
        """(square-root mismatches over every float in [2^-60, 2^60], quotient mismatches over n_div operand pairs) of the pair-cost kernels'
        instruction-level sqrt / divide against the device's IEEE results."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._chk(self._L.se3d_selftest_exact_math(C.c_void_p(self._h), C.c_uint64(int(n_div)), C.byref(a), C.byref(b)), "se3d_selftest_exact_math")
        return int(a.value), int(b.value)

    def selftest_shared_quotient(self, n=1 << 33):
        """(mismatches, skipped pairs) of the large-rig pair kernel's three-instruction quotient on the shared reciprocal against the device's
        IEEE quotient over n operand pairs of the kernel's guarded range."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._chk(self._L.se3d_selftest_shared_quotient(C.c_void_p(self._h), C.c_uint64(int(n)), C.byref(a), C.byref(b)), "se3d_selftest_shared_quotient")
        return int(a.value), int(b.value)

    def sweep_stats(self):
        """(camera steps of the per-camera sweep that met hypotheses and detections, those that ran the Hungarian): running totals (large rigs)."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        self._chk(self._L.se3d_get_sweep_stats(C.c_void_p(self._h), C.byref(a), C.byref(b)), "se3d_get_sweep_stats")
        return int(a.value), int(b.value)

    def pair_chunks(self):
        """Camera-column chunks of the pair table in this context (0: the whole table is materialised)."""
        return int(self._L.se3d_get_pair_chunks(C.c_void_p(self._h)))

    def selftest_dlt32(self, Rt, pts, weight_by_conf=True):
        """The second pass's float32 DLT solve (se3d_exact32.h) on caller systems: Rt [n_sys][n][12], pts [n_sys][n][3] (x, y, conf)
        -> (X [n_sys][3] float32, reprojection error [n_sys] float64)."""
        Rt = np.ascontiguousarray(Rt, np.float32)
        pts = np.ascontiguousarray(pts, np.float32)
        n_sys, n = pts.shape[0], pts.shape[1]
        assert Rt.shape == (n_sys, n, 12) and pts.shape == (n_sys, n, 3)
        X = np.zeros((n_sys, 3), np.float32)
        err = np.zeros(n_sys, np.float64)
        self._chk(self._L.se3d_selftest_dlt32(C.c_void_p(self._h), n_sys, n, pm.fptr(Rt), pm.fptr(pts), int(bool(weight_by_conf)), pm.fptr(X),
                                              err.ctypes.data_as(C.POINTER(C.c_double))), "se3d_selftest_dlt32")
        return X, err

    def exact_list(self, flags=False):
        """Joints the last triangulation pass handed to the float32-exact second pass: (array [n][3] of (frame, hypothesis, joint), running total).
        flags=True: (array, cov64 [n] bool, total) — cov64: the joint's covariance came from fp64 sigma samples around the float32 point (bit 31 of
        the entry; options.exact_cov), position, score and outlier decisions from the float32 sequence like every listed joint."""
        n, total = C.c_int32(0), C.c_int64(0)
        cap = self.limits.max_frames * self.limits.max_hyp * pm.NUM_KEYPOINTS
        buf = np.zeros(cap, np.uint32)
        self._chk(self._L.se3d_get_exact_list(C.c_void_p(self._h), buf.ctypes.data_as(C.POINTER(C.c_uint32)), cap, C.byref(n), C.byref(total)), "se3d_get_exact_list")
        e = buf[:min(n.value, cap)]
        fh = (e & 0x3fffffff) >> 5
        out = np.stack([fh // self.limits.max_hyp, fh % self.limits.max_hyp, e & 31], -1).astype(np.int64) if len(e) else np.zeros((0, 3), np.int64)
        if flags:
            return out, (e >> 31).astype(bool), int(total.value)
        return out, int(total.value)

    def profile(self, on=True):
        self._chk(self._L.se3d_profile_enable(C.c_void_p(self._h), int(on)), "se3d_profile_enable")

    def profile_reset(self):
        self._chk(self._L.se3d_profile_reset(C.c_void_p(self._h)), "se3d_profile_reset")

    def profile_get(self, name):
        ms, n = C.c_double(0), C.c_int64(0)
        self._chk(self._L.se3d_profile_get(C.c_void_p(self._h), name.encode(), C.byref(ms), C.byref(n)), "se3d_profile_get")
        return ms.value, n.value

    def close(self):
        if getattr(self, "_h", None):
            self._L.se3d_destroy(C.c_void_p(self._h))
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
