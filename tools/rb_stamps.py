#!/usr/bin/env python3
"""Diagnostic (make -C smartedgesensor3dhumanpose_amd/csrc diag): where a wavefront of se3d_k_pairs_rows spends its cycles at 256cam x 128p — wavefront 0 (stages
operands) and wavefront 3 (does not) of every 16th workgroup, stamped with the cycle counter.  Not a benchmark (the stamps add waits)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smartedgesensor3dhumanpose_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "tmp", "libse3d_hip_diag.so")
import bench
w = bench.WORKLOADS["256cam128p"]
rig = bench.make_rig(w)
T_ = 8
seqs = bench.make_sequences(None, w, 1, 1, 0, T_)[0]
frames = [seqs[0][k] for k in range(T_)]
from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
node = Skeleton3D(rig, max_epi_dist=0.045, max_frames=T_, max_det_per_cam=w["max_det"], max_hyp=w["max_hyp"])
node.stage(frames)
L = _lib.lib()
out = (C.c_ulonglong * 32)()
node.run_stage1(); node.sync(); L.se3d_diag_rb_stamps(out, 1)
node.run_stage1(); node.sync(); L.se3d_diag_rb_stamps(out, 1)
names = ["entry loads + barrier", "det_a operands -> LDS", "first det_b phase", "flag read", "stage next det_b", "flush + reset", "prefetch + det_b -> regs", "ITEMS", "barrier wait", "last flush"]
for wv, lab in ((0, "wavefront 0 (stages)"), (1, "wavefront 3 (no staging)")):
    v = list(out)[16 * wv:16 * wv + 16]
    nwg, nph = max(v[15], 1), max(v[14], 1)
    tot = sum(v[:10])
    print("%s: %d workgroups stamped, %.1f phases per workgroup, %.0f cycles per workgroup" % (lab, nwg, nph / nwg, tot / nwg))
    for i, nm in enumerate(names):
        print("   %-28s %8.0f cycles per workgroup (%5.1f %%)%s" % (nm, v[i] / nwg, 100.0 * v[i] / tot, "   = %.0f per phase" % (v[i] / nph) if 3 <= i <= 8 else ""))
