#!/usr/bin/env python3
"""Diagnostic (GPU box): where does a stage-2 sequence of the randomised sweep leave the oracle's?

  python tools/stage2_track_tables.py <seed> <trial> <sequence>

Replays one sequence of tests/fuzz_scenes.py frame by frame through se3d_triangulate + se3d_refine and through the oracle, and prints after
every step the track tables of both sides — (id, num_obs) from se3d_refine_save / PosePrior.tracks() — the oracle's detection -> track ids and
merges, and, at the first step whose tables differ, the track-association cost matrix (calc_normed_dist, PP:84-101) of that step evaluated on
the oracle's and on the library's track state.  This is how DESIGN section 4's statement "a badly scaled person contaminates the whole
sequence through the Hungarian" was established (seed 4, trial 2, sequence 2: profiles/r4/stage2_track_tables_seed4.txt)."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle_py as O
O.build()
from smartedgesensor3dhumanpose_amd import person_msgs as pm
from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
from fuzz_scenes import MAX_DET, MAX_HYP, N_SEQ, T_STEPS, draw_trial
from parity import garbage_persons
seed, want_trial, s = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
rng = np.random.default_rng(seed)
for trial in range(want_trial + 1):
    rig, persons, seqs, extra = draw_trial(rng)
frames = [seqs[s][k] for k in range(T_STEPS)]
node = Skeleton3D(rig, max_epi_dist=0.045, max_frames=1, max_det_per_cam=MAX_DET, max_hyp=MAX_HYP, device=0)
ora = O.PosePrior(node.params)
node.refine_reset()
from smartedgesensor3dhumanpose_amd.pose_prior import PosePrior
L = node._L
pp_gpu = PosePrior(node, 0)
def gpu_tracks():
    return {int(r["id"]): dict(num_obs=int(r["num_obs"]), mask=int(r["mask"]), t_prev=float(r["t_prev"]), height_prev=float(r["height_prev"]),
                               root_prev=r["root_prev"].copy(), prev=r["prev"].copy()) for r in pp_gpu.tracks()}
def cost(det, tr, t):   # calc_normed_dist (PP:84-101)
    dt = t - tr["t_prev"]; n = 0; dist = 0.0
    for k in range(21):
        if det["keypoints"]["score"][k] > 0.10 and (int(tr["mask"]) >> k) & 1:
            pk = tr["prev"][k] * tr["height_prev"] + tr["root_prev"]
            dist += np.linalg.norm(det["keypoints"]["joint"][k] - pk) / (vs[k] * dt); n += 1
    return dist / n if n else 1e6
idx22 = (C.c_int32 * 22)(); vs21 = (C.c_double * 21)()
L.se3d_get_fusion_constants(C.c_void_p(node._h), idx22, vs21)
vs = np.array(list(vs21))
for k, fr in enumerate(frames):
    t = 100.0 + (k + 1) / 30.0
    dets = node.triangulate_persons([fr])[0].published()
    g_before = gpu_tracks() if k > 0 else {}
    ot_b = ora.tracks() if k > 0 else None
    o2 = ora.step(dets, rig.n_cam, t, fr.fb_delay, capacity=MAX_HYP)
    g2 = node.refine(0, dets, t, fr.fb_delay)
    gi = gpu_tracks()
    ot = ora.tracks()
    gt = sorted((a, v["num_obs"]) for a, v in gi.items()); oo = sorted(zip(ot["ids"].tolist(), ot["num_obs"].tolist()))
    print("step", k, "garbage", np.nonzero(garbage_persons(dets))[0].tolist(), "det ids", ora.last_det_ids().tolist(), "merges", ora.last_merges())
    print("   ora tracks (id, obs):", oo)
    print("   gpu tracks (id, obs):", gt, "" if gt == oo else "  <-- DIFFERENT")
    for i_, id_ in enumerate(ot["ids"].tolist()):   # state differences of the tracks both sides have
        if id_ in gi:
            gtr = gi[id_]
            dr = np.linalg.norm(gtr["root_prev"] - ot["root_prev"][i_]); dp = np.abs(gtr["prev"] - ot["prev"][i_]).max(); dh = abs(gtr["height_prev"] - ot["height_prev"][i_])
            if max(dr, dp, dh) > 1e-6 or gtr["mask"] != int(ot["mask"][i_]):
                jw = int(np.argmax(np.abs(gtr["prev"] - ot["prev"][i_]).max(-1)))
                print("   track %d: root differs by %.3e m, height by %.3e, normalised joints by %.3e (joint %d: oracle %s, library %s), masks %x / %x" % (
                    id_, dr, dh, dp, jw, np.round(ot["prev"][i_][jw], 4).tolist(), np.round(gtr["prev"][jw], 4).tolist(), int(ot["mask"][i_]), gtr["mask"]))
    if gt != oo and k > 0 and not globals().get("shown"):
        shown = True
        print("   cost matrices BEFORE this step (rows: detections, columns: track ids), oracle state / gpu state:")
        ids = ot_b["ids"].tolist()
        for p in range(len(dets)):
            co = [cost(dets[p], dict(t_prev=ot_b["t_prev"][i], height_prev=ot_b["height_prev"][i], root_prev=ot_b["root_prev"][i], mask=ot_b["mask"][i], prev=ot_b["prev"][i]), t) for i in range(len(ids))]
            cg = [cost(dets[p], g_before[i_], t) if i_ in g_before else float("nan") for i_ in ids]
            print("     det", p, " ".join("%d:%.4f/%.4f" % (i_, a, b) for i_, a, b in zip(ids, co, cg) if a < 50 or b < 50))
