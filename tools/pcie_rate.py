#!/usr/bin/env python3
"""PCIe-inclusive rate of the hot path: every step stages the raw Person2D payloads from host memory
(se3d_stage_frames: host -> pinned -> HBM), runs stage 1 + the stage-2 walk and fetches all fused / predicted persons back to host
buffers (se3d_fetch_stage2_walk), against the resident rate bench.py reports.  One context, 16cam6p."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as Bn  # noqa: E402
from smartedgesensor3dhumanpose_amd import person_msgs as pm  # noqa: E402
from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D  # noqa: E402

w = Bn.WORKLOADS["16cam6p"]
rig = Bn.make_rig(w)
S_, T_ = w["seqs"], Bn.T_STEPS
B = S_ * T_
seqs = Bn.make_sequences(rig, w, S_, 0, T_)
frames = [seqs[s][k] for k in range(T_) for s in range(S_)]
node = Skeleton3D(rig, max_epi_dist=0.045, max_frames=B, max_det_per_cam=w["max_det"], max_hyp=w["max_hyp"])
L, h = node._L, C.c_void_p(node._h)
arr = (pm.FrameIn * B)(*[f.as_c() for f in frames])
res = [pm.FusedResult(w["max_hyp"]) for _ in range(B)]
outs = (pm.FusedOut * B)(*[r.c for r in res])
node.stage(frames)
node.run_stage1()
node.refine_reset()
t = 100.0


def step(pcie):
    global t
    if pcie:
        assert L.se3d_stage_frames(h, B, arr) == 0
    node.run_stage1()
    ts = t + np.arange(1, T_ + 1) / 30.0
    t = float(ts[-1])
    node.run_stage2_walk(np.repeat(ts[:, None], S_, 1), [k * S_ for k in range(T_)])
    if pcie:
        assert L.se3d_fetch_stage2_walk(h, 0, B, outs) == 0
        return sum(o.n_persons for o in outs)
    node.sync()
    return None


for mode in (False, True):
    for _ in range(3):
        step(mode)
    node.sync()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        pub = step(mode)
    node.sync()
    dt = (time.perf_counter() - t0) / n
    raw_mb = B * rig.n_cam * w["max_det"] * pm.person2d_dt.itemsize / 1e6
    out_mb = 2 * B * w["max_hyp"] * pm.person3d_dt.itemsize / 1e6
    print("%s: %.2f ms/step, %.0f frames/s%s" % ("PCIe-inclusive (stage %.0f MB in, fetch %.0f MB out per step)" % (raw_mb, out_mb) if mode else "resident", dt * 1e3, B / dt,
                                                  ", %.0f skeletons/s" % (pub / dt) if pub else ""))
