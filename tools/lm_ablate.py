#!/usr/bin/env python3
"""Diagnostic (make -C .../csrc diag): wall time of the fused stage-2 walk at 16cam6p with parts of the LM switched off, to attribute
the walk time without in-kernel stamps (which perturb memory phases).  Results with ablation bits set are wrong by construction.
The switches live in the one-person-per-wavefront LM (s2_lm_body), so the tool runs the walk with options.walk_lm = solo."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smartedgesensor3dhumanpose_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "tools", "tmp", "libse3d_hip_diag.so")
import numpy as np
import bench
w = bench.WORKLOADS["16cam6p"]
rig = bench.make_rig(w)
S_, T_ = 128, 16
seqs = bench.make_sequences(None, w, 1, S_, 0, T_)[0]
frames = [seqs[s][k] for k in range(T_) for s in range(S_)]
from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
node = Skeleton3D(rig, max_epi_dist=0.045, max_frames=S_ * T_, max_det_per_cam=8, max_hyp=12, options=dict(walk_lm="solo"))
node.stage(frames); node.refine_reset(); node.run_stage1(); node.sync()
L = _lib.lib()
t = [100.0]
def walk(rep):
    ks = list(range(T_)) if rep % 2 == 0 else list(range(T_ - 1, -1, -1))
    ts = t[0] + np.arange(1, T_ + 1) / 30.0
    t[0] = float(ts[-1])
    node.run_stage2_walk(np.repeat(ts[:, None], S_, 1), [k * S_ for k in ks])
for bits, name in ((0, "full"), (2, "no marginals"), (1, "one LM iteration"), (1 + 8, "one iteration, no solve"), (1 + 16, "one iteration, no linearize"),
                   (1 + 32, "one iteration, no linear error"), (1 + 64, "one iteration, no new error"), (1 + 8 + 16 + 32 + 64, "one iteration, loop control only"),
                   (4, "no LM loop"), (6, "no LM loop, no marginals")):
    L.se3d_diag_lm_ablate(bits)
    for r in range(4): walk(r)
    node.sync()
    t0 = time.perf_counter()
    for r in range(10): walk(r)
    node.sync()
    print("%-28s %.3f ms per walk (128 sequences x 16 steps)" % (name, (time.perf_counter() - t0) / 10 * 1e3), flush=True)
