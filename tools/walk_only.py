"""Diagnostic: throughput of the fused stage-2 walk alone (no stage 1 in the loop) for C contexts:  python tools/walk_only.py C"""
import sys, os, time, numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
sys.path.insert(0, os.getcwd())
import bench as Bn
from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
w = Bn.WORKLOADS["16cam6p"]
rig = Bn.make_rig(w)
S_, T_, NC = (int(sys.argv[2]) if len(sys.argv) > 2 else 128), 16, int(sys.argv[1])
nodes = []
for ci in range(NC):
    seqs = Bn.make_sequences(None, w, 1, S_, ci, T_)[0]
    frames = [seqs[s][k] for k in range(T_) for s in range(S_)]
    nd = Skeleton3D(rig, max_epi_dist=0.045, max_frames=S_ * T_, max_det_per_cam=8, max_hyp=12)
    nd.stage(frames); nd.run_stage1(); nd.refine_reset(); nodes.append(nd)
t = [100.0] * NC
cyc = [0] * NC
def walk(ci):
    ks = list(range(T_)) if cyc[ci] % 2 == 0 else list(range(T_ - 1, -1, -1))
    ts = t[ci] + np.arange(1, T_ + 1) / 30.0
    t[ci] = float(ts[-1]); cyc[ci] += 1
    nodes[ci].run_stage2_walk(np.repeat(ts[:, None], S_, 1), [k * S_ for k in ks])
for _ in range(3):
    for ci in range(NC): walk(ci)
for nd in nodes: nd.sync()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    for ci in range(NC): walk(ci)
for nd in nodes: nd.sync()
dt = (time.perf_counter() - t0) / n
print("contexts", NC, "walks only: %.3f ms per step (%.3f per context)" % (dt * 1e3, dt * 1e3 / NC))
