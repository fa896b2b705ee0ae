#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of se3d_k_lm_refine from the stamp build (make -C .../csrc diag).  Not a benchmark."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smartedgesensor3dhumanpose_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "tmp", "libse3d_hip_diag.so")
import numpy as np
import bench
w = bench.WORKLOADS["16cam6p"]
rig = bench.make_rig(w)
S_, T_ = int(os.environ.get("STAMP_SEQS", "128")), 16
seqs = bench.make_sequences(None, w, 1, S_, 0, T_)[0]
frames = [seqs[s][k] for k in range(T_) for s in range(S_)]
from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
node = Skeleton3D(rig, max_epi_dist=0.045, max_frames=S_ * T_, max_det_per_cam=8, max_hyp=8)
node.stage(frames); node.refine_reset(); node.run_stage1()
L = _lib.lib()
out = (C.c_ulonglong * 16)()
t = 100.0
for rep in range(3):
    for k in (range(T_) if rep % 2 == 0 else range(T_ - 1, -1, -1)):
        t += 1 / 30
        node.run_stage2(np.full(S_, t), frame_slot0=k * S_)
    node.sync()
    L.se3d_diag_lm_stamps(out, 1)
    v = list(out)
    n = max(v[8], 1)
    names = ["setup", "error/ctl", "linearize+grad", "solve", "lm-rest", "marginals", "outputs"]
    print("rep", rep, "persons", v[8], "avg LM iterations %.2f" % (v[9] / n), "| cycles/person:", {nm: round(v[i] / n) for i, nm in enumerate(names)}, "total", round(sum(v[:7]) / n), "| outputs split: zero-fill", round(v[7] / n), "scalar loads", round(v[15] / n))

# fused walk (se3d_run_stage2_walk): per-phase cycles of thread 0 of each sequence's workgroup, per (sequence, time step)
L.se3d_diag_lm_stamps(out, 1)
for rep in range(2):
    ks = list(range(T_)) if rep % 2 == 0 else list(range(T_ - 1, -1, -1))
    ts = t + np.arange(1, T_ + 1) / 30.0
    t = float(ts[-1])
    node.run_stage2_walk(np.repeat(ts[:, None], S_, 1), [k * S_ for k in ks])
    node.sync()
    L.se3d_diag_lm_stamps(out, 1)
    v = list(out)
    n = S_ * T_
    if v[8]:
        nt = v[8]
        tn = ["setup", "linearize+grad", "solve", "errors/ctl", "-", "marginals", "outputs"]
        print("  trios (thread 0's wavefront)", nt, "avg trips %.2f" % (v[9] / nt), "| cycles/trio:", {nm: round(v[i] / nt) for i, nm in enumerate(tn) if nm != "-"},
              "| per trip:", {nm: round(v[i] / max(v[9], 1)) for i, nm in enumerate(tn) if nm in ("linearize+grad", "solve", "errors/ctl")})
    names = ["track_cost", "assoc", "lm(max of seq)", "near", "update"]
    # the diagnostic build stamps one workgroup in 64 (uncontended atomics): scale back to per (sequence, step)
    print("walk rep", rep, "| cycles per (sequence, step):", {nm: round(64 * v[10 + i] / n) for i, nm in enumerate(names)}, "total", round(64 * sum(v[10:15]) / n))
