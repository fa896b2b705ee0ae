#!/usr/bin/env python3
"""Diagnostic: per-phase cycles of se3d_k_triangulate_hyp from the stamp build (make -C .../csrc diag; one workgroup in 64 is stamped).
Not a benchmark.  STAMP_SEQS sequences x 16 frames of the 16cam6p workload in one context."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smartedgesensor3dhumanpose_amd import _lib
_lib.LIB_PATH = os.environ.get("STAMP_LIB") or os.path.join(ROOT, "tools", "tmp", "libse3d_hip_diag.so")
import bench
w = bench.WORKLOADS["16cam6p"]
rig = bench.make_rig(w)
S_, T_ = int(os.environ.get("STAMP_SEQS", "2048")), 16
seqs = bench.make_sequences(None, w, 1, min(S_, 32), 0, T_)[0]
frames = [seqs[s % len(seqs)][k] for k in range(T_) for s in range(S_)]
from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
node = Skeleton3D(rig, max_epi_dist=0.045, max_frames=S_ * T_, max_det_per_cam=8, max_hyp=8)
node.stage(frames)
L = _lib.lib()
out = (C.c_ulonglong * 16)()
names = ["entry loads", "camera list + unused slots", "points + rows + view lists", "-", "Gram", "weighted solve", "reproj + score", "writes", "sigma samples", "cov sums + writes"]
for rep in range(3):
    node.run_stage1(); node.sync()
    L.se3d_diag_th_stamps(out, 1)
    v = list(out); n = max(v[15], 1)
    print("rep", rep, "workgroups stamped", v[15], "| cycles per workgroup:", {nm: round(v[i] / n) for i, nm in enumerate(names)}, "total", round(sum(v[:10]) / n))

import numpy as np
res = node.triangulate_persons(frames[:256])
nv = np.concatenate([(r.assoc() >= 0).sum(1) for r in res])
print("views per hypothesis (256 frames): histogram", np.bincount(nv, minlength=17).tolist(), "mean", round(float(nv.mean()), 2))
