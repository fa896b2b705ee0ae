#!/usr/bin/env python3
"""Diagnostic: cycle shares of se3d_k_front_fused (one 16cam6p frame per call) from the stamp build (make -C .../csrc diag).  Not a benchmark."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smartedgesensor3dhumanpose_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "tmp", "libse3d_hip_diag.so")
import bench
w = bench.WORKLOADS["16cam6p"]
rig = bench.make_rig(w)
seqs = bench.make_sequences(None, w, 1, 1, 0, 16)[0]
frames = [seqs[0][k] for k in range(16)]
from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
os.environ["SE3D_GRAPH"] = "0"
node = Skeleton3D(rig, max_epi_dist=0.045, max_frames=1, max_det_per_cam=w["max_det"], max_hyp=w["max_hyp"])
L = _lib.lib()
out = (C.c_ulonglong * 16)()
for fr in frames[:4]:
    node.triangulate_persons([fr])
L.se3d_diag_ff_stamps(out, 1)
for fr in frames:
    node.triangulate_persons([fr])
L.se3d_diag_ff_stamps(out, 1)
v = [x / len(frames) for x in out]
names = ["prologue", "prefix", "pair costs", "sweep (total, wave 0)", "  cost + veto mask", "  scans + Hungarian", "  attach / spawn", "  loop / seeding"]
for n, x in zip(names, v):
    print("%-26s %9.0f cycles per frame" % (n, x))
