"""Randomised parity sweep, shared by tests/test_gpu_fuzz_parity.py (asserts) and tools/fuzz_parity.py (prints): the default HIP kernels
through the C-ABI against the CPU oracle on the scenes of tests/fuzz_scenes.py — stage 1 per frame, the resident stage-2 walk per sequence.

What must hold (round 4: no exclusion set for stage-1 positions and scores any more — joints the fp64 fast path cannot guarantee are redone by
the library itself with the oracle's restatement of the reference's float32 arithmetic, csrc/se3d_exact32.h):
  * stage-1 index results (flags, hypothesis counts, association tables, published persons, sets of valid joints): identical in every frame
    (frames on which the library raised FLAG_HYP_OVERFLOW are counted; the oracle, run with room to spare, must have needed more);
  * stage-1 positions < 1 mm and scores bit-equal (SCORE_RTOL = 0) on EVERY valid joint;
  * the library's own flag is checked against the inputs: every published joint whose DLT systems have an input-derived kappa >=
    DLT_KAPPA_MAX (tests/parity.py, fp64 singular values) must be among the joints the library sent through its float32 second pass
    (se3d_get_exact_list) — counted as `ill_not_listed`, asserted zero for unmerged persons;
  * stage-1 covariances: relative Frobenius < 1e-3 + 2 2^-24 min(kappa, DLT_KAPPA_MAX) / sigma on every valid joint with a finite covariance
    (parity.cov_tolerance: the float32 noise of the reference's sample solves on the fast path's joints; second-pass joints carry the
    reference's bits), after the allowance for what the measured position difference does to the limb-length term (parity.inflation_allowance);
  * stage-2 ids / counts / LM iteration counts identical, refined joints < 1 mm, marginal covariances rtol 1e-4 on every sequence-step of a
    sequence that has not yet seen a garbage person (parity.garbage_persons); from the first such person on the sequence is in the exclusion
    set (counted), because its LM problem is solver-dependent and the difference lives on in the track state;
  * round 5, on EVERY sequence-step (exempt or not): a published person whose refined joints differ by 1 mm or more between the two sides must
    not have the worse final LM objective on the library's side (parity.compare_stage2_objectives; steps on which the two sides publish
    different detections are counted as not comparable), and the mismatches INSIDE the exclusion set are bounded by the callers;
  * round 6: those mismatches are counted by KIND (parity.classify_stage2) — a different person / track count or id is an index divergence and has
    its own, tighter bound; a predicted joint >= 1 mm apart under refined joints that agree (a velocity-history difference) its own counter."""
import os

import numpy as np

from smartedgesensor3dhumanpose_amd import person_msgs as pm
from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
from fuzz_scenes import MAX_DET, MAX_HYP, N_SEQ, T_STEPS, draw_trial
from parity import COV_REL_FRO, DLT_KAPPA_MAX, POS_TOL_M, SCORE_RTOL, classify_stage2, compare_stage2, compare_stage2_objectives, cov_tolerance, garbage_persons, inflation_allowance, published_joint_condition

W6 = np.array([1, 2, 2, 1, 2, 1.0])


def new_report(seed):
    return dict(seed=seed, trials=0, frames=0, frames_hyp_overflow=0, persons=0, joints=0, joints_excluded=0, joints_second_pass=0, ill_joints=0, ill_not_listed=0,
                joints_over_1mm=0, pos_max_mm=0.0, scores_over_rtol=0, cov_over_tol=0, cov_rel_max=0.0, cov_nan=0, **{"cov_over_flat_1e-3": 0},
                seq_steps=0, seq_steps_track_overflow=0, seq_steps_excluded=0, stage2_numeric_failures=0, stage2_index_divergences=0,
                excluded_stage2_mismatches=0, excluded_count_mismatches=0, excluded_pred_only_mismatches=0, excluded_mismatch_kinds={}, objective_pred_only=0, objective_persons=0, objective_disagree=0, objective_worse=0, objective_worst_excess=0.0, objective_not_comparable_steps=0,
                violations=[])


def dlt_condition_own(rig, frame, result):
    """kappa [n_persons][17] of the published persons' OWN hypotheses (weighted and unweighted system, no leave-one-out subsets, no merge
    partners): what the library's in-kernel flag (se3d_kappa_flag) bounds from above."""
    from parity import plain_condition
    n = result.n_persons
    return plain_condition(rig, frame, result.assoc())[result.person_hyp[:n]]


def run_trials(O, seed, trials, walk_mode="wave", pose_method="simple", norm_height=False):
    """O: the oracle module (tests' fixture / oracle_py).  Index mismatches of stage 1 raise at once; everything else is tallied.
    pose_method / norm_height: the nodes' ROS parameters (S3D:1095, PP:930-931) — "h36m" keypoint set, height-normalised LM."""
    rng = np.random.default_rng(seed)
    rep = new_report(seed)
    rep["pose_method"], rep["norm_height"] = pose_method, bool(norm_height)
    tables = O.pose_tables(pm.POSE_H36M if pose_method == "h36m" else pm.POSE_SIMPLE)
    k2f = np.array(tables["kp2fusion"])
    if pose_method == "h36m":
        tables = dict(tables, shoulderDist=0.0)   # the shoulder special case exists in the "simple" model only (S3D:875)
    for trial in range(trials):
        rig, persons, seqs, extra = draw_trial(rng)
        frames = [seqs[s][k] for k in range(T_STEPS) for s in range(N_SEQ)]
        node = Skeleton3D(rig, pose_method=pose_method, norm_height=norm_height, max_epi_dist=0.045, max_frames=N_SEQ * T_STEPS, max_det_per_cam=MAX_DET,
                          max_hyp=MAX_HYP, device=0)
        params = node.params
        node.stage(frames)
        node.run_stage1()
        node.refine_reset()
        r1 = node.fetch_stage1(N_SEQ * T_STEPS)
        listed, _ = node.exact_list()   # (frame, hypothesis, joint) of the library's float32 second pass
        rep["joints_second_pass"] += len(listed)
        for fi, fr in enumerate(frames):
            g, o = r1[fi], O.triangulate(rig, params, fr, capacity=4 * MAX_HYP, hyp_capacity=4 * MAX_HYP)   # the oracle with room to spare
            rep["frames"] += 1
            if g.flags & pm.FLAG_HYP_OVERFLOW:
                assert o.n_hyp > MAX_HYP, ("overflow flag without overflow", seed, trial, fi, o.n_hyp)
                rep["frames_hyp_overflow"] += 1
                continue
            assert g.flags == o.flags and o.n_hyp <= MAX_HYP, ("flags", seed, trial, fi, g.flags, o.flags, o.n_hyp, "library: hypotheses, persons", g.n_hyp, g.n_persons,
                                                               "oracle persons", o.n_persons, "cameras with detections", int((np.diff(np.asarray(fr.det_offset)) > 0).sum()))
            assert g.n_hyp == o.n_hyp, ("hypotheses", seed, trial, fi, g.n_hyp, o.n_hyp)
            assert np.array_equal(g.assoc(), o.assoc()), ("association table", seed, trial, fi)
            assert g.n_persons == o.n_persons and np.array_equal(g.person_hyp[:g.n_persons], o.person_hyp[:g.n_persons]), ("published persons", seed, trial, fi)
            n = g.n_persons
            if n == 0:
                continue
            gk, ok = g.published()["keypoints"][:, k2f], o.published()["keypoints"][:, k2f]     # the 17 triangulated joints, input order
            v = ok["score"] > 0
            assert np.array_equal(gk["score"] > 0, v), ("valid joints", seed, trial, fi)
            kappa = published_joint_condition(rig, fr, g, k2f)
            assert not np.isnan(kappa[v]).any(), ("published joint with fewer than two usable views", seed, trial, fi)
            # the library's own conditioning flag against the inputs: an ill-conditioned joint of a person's OWN hypothesis must be listed
            # (kappa of published_joint_condition also covers possible merge partners, so the own-hypothesis figure is taken again here)
            own = dlt_condition_own(rig, fr, g)
            lf = listed[listed[:, 0] == fi]
            is_listed = np.zeros((n, pm.NUM_KEYPOINTS), bool)
            row_of = {int(h): i for i, h in enumerate(g.person_hyp[:n])}
            for _, h, j in lf:
                if int(h) in row_of:
                    is_listed[row_of[int(h)], j] = True
            ill = v & (own >= DLT_KAPPA_MAX)
            rep["ill_joints"] += int(ill.sum())
            rep["ill_not_listed"] += int((ill & ~is_listed).sum())
            d = np.linalg.norm(gk["joint"] - ok["joint"], axis=-1)
            with np.errstate(invalid="ignore", divide="ignore"):
                rs = np.abs(gk["score"] - ok["score"]) / ok["score"]
                cg, co = gk["cov"], ok["cov"]
                fro_o = np.sqrt((co ** 2 * W6).sum(-1))
                # what the position difference itself (inside its tolerance) does to the limb-length term of the covariance: parity.inflation_allowance
                rel = np.maximum(np.sqrt(((cg - co) ** 2 * W6).sum(-1)) - inflation_allowance(tables, ok["joint"], v, np.where(v, d, 0.0)), 0.0) / fro_o
            nan_o, nan_g = np.isnan(co).any(-1) & v, np.isnan(cg).any(-1) & v
            assert np.array_equal(nan_g, nan_o), ("NaN covariances", seed, trial, fi)
            rep["cov_nan"] += int(nan_o.sum())
            relv = np.where(nan_o, 0.0, rel)
            rep["persons"] += n
            rep["joints"] += int(v.sum())
            rep["joints_over_1mm"] += int((d[v] >= POS_TOL_M).sum())
            rep["pos_max_mm"] = max(rep["pos_max_mm"], float(d[v].max() * 1e3))
            rep["scores_over_rtol"] += int((rs[v] > SCORE_RTOL).sum())
            ctol = cov_tolerance(np.minimum(kappa, DLT_KAPPA_MAX), co)
            vc = v & ~nan_o
            rep["cov_over_tol"] += int((relv[vc] >= ctol[vc]).sum())
            rep["cov_over_flat_1e-3"] += int((relv[vc] >= COV_REL_FRO).sum())
            rep["cov_rel_max"] = max(rep["cov_rel_max"], float(relv[vc].max()) if vc.any() else 0.0)
            for p, k in zip(*np.nonzero(v & ((d >= POS_TOL_M) | (vc & (relv >= ctol)) | (rs > SCORE_RTOL)))):
                rep["violations"].append(dict(trial=trial, frame=fi, person=int(p), joint=int(k), kappa=float(kappa[p, k]), d_mm=float(d[p, k] * 1e3),
                                              cov_rel=float(relv[p, k]), score_rel=float(rs[p, k]), listed=bool(is_listed[p, k])))
        # ---- stage 2: the resident walk; ids / counts / iteration counts exact inside compare_stage2 ----
        node.set_options(**(dict(walk_shape=walk_mode) if walk_mode in ("wave", "wg") else dict(walk_form="steps")))
        ts = [100.0 + (i + 1) / 30.0 for i in range(T_STEPS)]
        node.run_stage2_walk(np.repeat(np.array(ts)[:, None], N_SEQ, 1), [k * N_SEQ for k in range(T_STEPS)])
        node.sync()
        oras = [O.PosePrior(params) for _ in range(N_SEQ)]
        dead, tainted = [False] * N_SEQ, [False] * N_SEQ
        for k, t in enumerate(ts):
            ra = node.fetch_stage2_walk(k * N_SEQ, N_SEQ)
            for s in range(N_SEQ):
                if dead[s]:
                    continue
                rep["seq_steps"] += 1
                if ra[s].c.flags & pm.S2FLAG_TRACK_OVERFLOW:   # track capacity exceeded: truncated by contract, the oracle has no capacity
                    rep["seq_steps_track_overflow"] += 1
                    dead[s] = True
                    continue
                dets = r1[k * N_SEQ + s].published()
                tainted[s] = tainted[s] or bool(garbage_persons(dets).any())
                o2 = oras[s].step(dets, rig.n_cam, t, seqs[s][k].fb_delay, capacity=MAX_HYP)
                if tainted[s]:
                    rep["seq_steps_excluded"] += 1
                # EVERY sequence-step, exempt or not (round 5): where the two sides publish different joints, the library's optimum must not be the
                # worse one by the factor graph's own objective (parity.compare_stage2_objectives)
                oc = compare_stage2_objectives(ra[s], o2)
                if oc["comparable"]:
                    rep["objective_persons"] += oc["persons"]
                    rep["objective_disagree"] += oc["disagree"]
                    rep["objective_worse"] += oc["worse"]
                    rep["objective_pred_only"] += oc["pred_only"]
                    rep["objective_worst_excess"] = max(rep["objective_worst_excess"], oc["worst_excess"])
                    if oc["worse"]:
                        rep["violations"].append(dict(trial=trial, step=k, seq=s, objective_worse=oc["worse"], excess=oc["worst_excess"], tainted=tainted[s]))
                else:
                    rep["objective_not_comparable_steps"] += 1
                try:
                    compare_stage2(ra[s], o2, check_tracks=(k == T_STEPS - 1))
                except AssertionError as e:
                    msg = str(e)
                    numeric = "mm" in msg or "cov" in msg or "marg" in msg
                    if tainted[s]:
                        rep["excluded_stage2_mismatches"] += 1
                        kind = classify_stage2(ra[s], o2) or "other"   # round 6: by kind — a different person COUNT is an index divergence, not a numeric one
                        rep["excluded_mismatch_kinds"][kind] = rep["excluded_mismatch_kinds"].get(kind, 0) + 1
                        rep["excluded_count_mismatches"] += int(kind in ("count", "ids"))
                        rep["excluded_pred_only_mismatches"] += int(kind == "pred")
                        rep.setdefault("excluded_stage2_messages", []).append("trial %d step %d seq %d: %s" % (trial, k, s, msg[:160]))
                        dead[s] = dead[s] or not numeric
                        continue
                    rep["stage2_numeric_failures" if numeric else "stage2_index_divergences"] += 1
                    with np.errstate(invalid="ignore"):   # what the step's INPUT looked like: the widest measured joint (stage-1 sigma) and the farthest from its person
                        vv = dets["keypoints"]["score"] > 0
                        sig = np.sqrt(np.nanmax(np.where(vv[..., None], dets["keypoints"]["cov"][..., [0, 3, 5]], 0.0))) if len(dets) else 0.0
                    rep["violations"].append(dict(trial=trial, step=k, seq=s, stage2=msg[:200], widest_input_sigma_m=float(sig)))
                    dead[s] = dead[s] or not numeric
        node.close()
        rep["trials"] += 1
    rep["tolerances"] = {"POS_TOL_M": POS_TOL_M, "SCORE_RTOL": SCORE_RTOL, "COV_REL_FRO": COV_REL_FRO, "DLT_KAPPA_MAX": DLT_KAPPA_MAX}
    return rep
