import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd())
import bench as Bn
from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
w = Bn.WORKLOADS["16cam6p"]
rig = Bn.make_rig(w)
S_, T_, NC = 128, 16, int(sys.argv[1]) if len(sys.argv) > 1 else 3
nodes = []
for ci in range(NC):
    seqs = Bn.make_sequences(rig, w, S_, seed0=100 * ci, t_steps=T_)
    frames = [seqs[s][k] for k in range(T_) for s in range(S_)]
    nd = Skeleton3D(rig, max_epi_dist=0.045, max_frames=S_ * T_, max_det_per_cam=w["max_det"], max_hyp=w["max_hyp"])
    nd.stage(frames); nd.refine_reset(); nodes.append(nd)
t = [100.0]
def step():
    for nd in nodes: nd.run_stage1()
    for k in range(T_):
        t[0] += 1 / 30.
        tt = np.full(S_, t[0])
        for nd in nodes: nd.run_stage2(tt, frame_slot0=k * S_)
for _ in range(2): step()
for nd in nodes: nd.sync()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(5): step()
    t1 = time.perf_counter()
    for nd in nodes: nd.sync()
    t2 = time.perf_counter()
    print("contexts", NC, "enqueue ms/step", (t1 - t0) / 5 * 1e3, "total ms/step", (t2 - t0) / 5 * 1e3, flush=True)
