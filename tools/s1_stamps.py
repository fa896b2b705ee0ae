#!/usr/bin/env python3
"""Diagnostic: cycle shares of se3d_k_sweep_assign at 256cam x 128p from the stamp build.  Not a benchmark."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smartedgesensor3dhumanpose_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "tmp", "libse3d_hip_diag.so")
import bench
wl = sys.argv[1] if len(sys.argv) > 1 else "256cam128p"
w = bench.WORKLOADS[wl]
rig = bench.make_rig(w)
seqs = bench.make_sequences(None, w, 1, 1, 0, 4)[0]
frames = [seqs[0][k] for k in range(4)]
from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
node = Skeleton3D(rig, max_epi_dist=0.045, max_frames=4, max_det_per_cam=w["max_det"], max_hyp=w["max_hyp"])
node.stage(frames)
L = _lib.lib()
out = (C.c_ulonglong * 16)()
mk = (C.c_ulonglong * 16)()
node.run_stage1(); node.sync(); L.se3d_diag_s1_stamps(out, 1); L.se3d_diag_mk_stamps(mk, 1)
node.run_stage1(); node.sync(); L.se3d_diag_s1_stamps(out, 1); L.se3d_diag_mk_stamps(mk, 1)
m = list(mk); n = max(m[0], 1)
print("munkres calls %d | per call: reduce+masks %.0f cyc, serial %.0f cyc, step5 passes %.1f (%.0f cyc), step-3 events %.1f, augmentations %.1f, reload+init starring %.0f cyc"
      % (m[0], m[1] / n, m[2] / n, m[3] / n, m[4] / n, m[5] / n, m[6] / n, m[7] / n))
v = list(out)
print("assign blocks that reached the scan:", v[8], "ambiguous (Hungarian ran):", v[9])
print("cycles per block: scans %.0f, hungarian(avg over all) %.0f, hungarian(avg when run) %.0f, apply %.0f"
      % (v[0] / max(v[8], 1), v[1] / max(v[8], 1), v[1] / max(v[9], 1), v[2] / max(v[8], 1)))
r = node.fetch_stage1(4)
print("persons per frame", [x.n_persons for x in r], "hyps", [x.n_hyp for x in r])
