import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
import bench as Bn
from smartedgesensor3dhumanpose_amd import person_msgs as pm
from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
w = Bn.WORKLOADS["16cam6p"]
rig = Bn.make_rig(w)
S_, T_ = 128, 16
for ci in range(3):
    seqs = Bn.make_sequences(rig, w, S_, seed0=100 * ci, t_steps=T_)
    frames = [seqs[s][k] for k in range(T_) for s in range(S_)]
    nd = Skeleton3D(rig, max_epi_dist=0.045, max_frames=S_ * T_, max_det_per_cam=w["max_det"], max_hyp=w["max_hyp"])
    nd.stage(frames)
    nd.run_stage1(); nd.sync()
    print("ctx", ci, "stage1 done", flush=True)
    res = nd.fetch_stage1(S_ * T_)
    bad = 0
    for fi, r in enumerate(res):
        p = r.published()
        j = p["keypoints"]["joint"]; c = p["keypoints"]["cov"]; sc = p["keypoints"]["score"]
        if not (np.isfinite(j).all() and np.isfinite(c).all() and np.isfinite(sc).all()):
            bad += 1
            if bad < 5:
                k = np.argwhere(~np.isfinite(j).all(-1) | ~np.isfinite(c).all(-1))
                print("frame", fi, "nonfinite at", k[:4].tolist(), j[tuple(k[0])], c[tuple(k[0])], sc[tuple(k[0])], flush=True)
    print("ctx", ci, "frames with non-finite:", bad, "max |joint|", max(np.abs(r.published()["keypoints"]["joint"]).max() for r in res if r.n_persons), flush=True)
    nd.refine_reset()
    t = 100.0
    for k in range(T_):
        t += 1 / 30.
        nd.run_stage2(np.full(S_, t), frame_slot0=k * S_); nd.sync()
        print("stage2 step", k, "ok", flush=True)
    nd.close()
