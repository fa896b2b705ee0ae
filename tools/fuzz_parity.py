#!/usr/bin/env python3
"""Randomised parity sweep (GPU box): the default HIP kernels against the CPU oracle on random scenes — rigs of 16 / 24 cameras, 2-7 persons,
missing views, low-confidence joints, gross 2-D outliers, 0.5-4 px noise, people standing 0.5-1.2 m apart — stage 1 per frame and the
resident stage-2 walk per sequence.  Stage-1 index results (flags, hypothesis counts, association tables, published persons, valid-joint sets) must be EXACT and abort the run
otherwise; stage-2 index results (track ids / counts, LM iteration counts) are exact unless an ill-conditioned LM problem sends the
three-persons-per-wavefront LM down another accept / reject path than the oracle (counted as `stage2_index_divergences`); numeric results are tallied against the test tolerances
(tests/parity.py) instead of asserted, because rough scenes contain ill-conditioned joints (two usable views with nearly parallel rays)
on which the float32 SVD of the reference / oracle and the fp64 solve of the kernels pick different points of an almost flat minimum: same
reprojection error and score, positions up to metres apart.  Not a test (tests/ uses fixed scenes inside the tolerances); evidence for
DESIGN.md §4.   python tools/fuzz_parity.py [seed] [trials]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from smartedgesensor3dhumanpose_amd import person_msgs as pm, rig as R, scene as S   # noqa: E402
from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D, rig_tensors       # noqa: E402
from oracle import oracle_py as O                                                       # noqa: E402
from parity import COV_REL_FRO, POS_TOL_M, SCORE_RTOL, compare_stage2                  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = np.random.default_rng(seed)
t0 = time.time()
tally = dict(trials=0, frames=0, frames_hyp_overflow=0, persons=0, joints=0, joints_over_test_tol=0, joints_over_1mm=0, pos_max_mm=0.0,
             joints_over_1mm_with_equal_score=0, scores_over_rtol=0, cov_over_tol=0, cov_rel_max=0.0, cov_nan=0,
             seq_steps=0, seq_steps_track_overflow=0, stage2_numeric_failures=0, stage2_index_divergences=0)
w6 = np.array([1, 2, 2, 1, 2, 1.0])
for trial in range(trials):
    ncam = int(rng.choice([16, 16, 24]))
    rig = R.extr_rig_16cam() if ncam == 16 else R.elliptical_rig(24, seed=int(rng.integers(100)))
    rig.set_tensors(*rig_tensors(rig))
    persons = int(rng.integers(2, 8))
    T_, S_, mh = 12, 3, 16
    extra = dict(p_miss=float(rng.uniform(0, 0.15)), low_score_frac=float(rng.uniform(0, 0.3)), outlier_frac=float(rng.uniform(0, 0.012)),
                 pixel_noise=float(rng.uniform(0.5, 4.0)), spacing=float(rng.uniform(0.5, 1.2)))
    seqs = [S.generate_sequence(rig, S.SceneConfig(n_persons=persons, n_frames=T_, seed=int(rng.integers(1 << 30)), **extra))[0] for _ in range(S_)]
    frames = [seqs[s][k] for k in range(T_) for s in range(S_)]
    # FUZZ_VARIANTS=1: also the h36m keypoint set and the height-normalised LM (drawn after the scene, so the scenes of a seed stay the same)
    variants = bool(os.environ.get("FUZZ_VARIANTS"))
    pose = "h36m" if variants and trial % 3 == 1 else "simple"
    norm_h = variants and trial % 2 == 1
    node = Skeleton3D(rig, pose_method=pose, norm_height=norm_h, max_epi_dist=0.045, max_frames=S_ * T_, max_det_per_cam=12, max_hyp=mh, device=0)
    params = node.params
    node.stage(frames)
    node.run_stage1()
    node.refine_reset()
    r1 = node.fetch_stage1(S_ * T_)
    for fi, fr in enumerate(frames):
        g, o = r1[fi], O.triangulate(rig, params, fr, capacity=4 * mh, hyp_capacity=4 * mh)   # the oracle with room to spare
        tally["frames"] += 1
        # ---- index results: exact or abort ----
        if g.flags & pm.FLAG_HYP_OVERFLOW:   # more hypotheses than max_hyp: the kernels truncate and say so; the oracle must indeed have needed more
            assert o.n_hyp > mh, ("overflow flag without overflow", trial, fi, o.n_hyp)
            tally["frames_hyp_overflow"] += 1
            continue
        assert g.flags == o.flags and o.n_hyp <= mh, ("flags", trial, fi, g.flags, o.flags, o.n_hyp)
        assert g.n_hyp == o.n_hyp, ("hypotheses", trial, fi, g.n_hyp, o.n_hyp)
        assert np.array_equal(g.assoc(), o.assoc()), ("association table", trial, fi)
        assert g.n_persons == o.n_persons and np.array_equal(g.person_hyp[:g.n_persons], o.person_hyp[:g.n_persons]), ("published persons", trial, fi)
        if g.n_persons == 0:
            continue
        gk, ok = g.published()["keypoints"], o.published()["keypoints"]
        v = ok["score"] > 0
        assert np.array_equal(gk["score"] > 0, v), ("valid joints", trial, fi)
        # ---- numeric results: tallied ----
        d = np.linalg.norm(gk["joint"] - ok["joint"], axis=-1)[v]
        rs = np.abs(gk["score"][v] - ok["score"][v]) / ok["score"][v]
        tally["persons"] += g.n_persons
        tally["joints"] += int(v.sum())
        tally["joints_over_test_tol"] += int((d >= POS_TOL_M).sum())
        tally["joints_over_1mm"] += int((d >= 1e-3).sum())
        tally["joints_over_1mm_with_equal_score"] += int(((d >= 1e-3) & (rs < 1e-5)).sum())
        tally["pos_max_mm"] = max(tally["pos_max_mm"], float(d.max() * 1e3))
        tally["scores_over_rtol"] += int((rs > SCORE_RTOL).sum())
        cg, co = gk["cov"][v], ok["cov"][v]
        nan = np.isnan(co).any(-1)
        assert np.array_equal(np.isnan(cg).any(-1), nan), ("NaN covariances", trial, fi)
        tally["cov_nan"] += int(nan.sum())
        rel = np.sqrt(((cg[~nan] - co[~nan]) ** 2 * w6).sum(-1)) / np.sqrt((co[~nan] ** 2 * w6).sum(-1))
        tally["cov_over_tol"] += int((rel >= COV_REL_FRO).sum())
        tally["cov_rel_max"] = max(tally["cov_rel_max"], float(rel.max()) if len(rel) else 0.0)
    # ---- stage 2: the resident walk (one wavefront per sequence), track ids / counts / iteration counts exact inside compare_stage2 ----
    mode = os.environ.get("FUZZ_MODE", "wave")
    if mode in ("wave", "wg"):
        os.environ["SE3D_WALK_SHAPE"] = mode
    else:
        os.environ["SE3D_WALK"] = "steps"
    ts = [100.0 + (i + 1) / 30.0 for i in range(T_)]
    node.run_stage2_walk(np.repeat(np.array(ts)[:, None], S_, 1), [k * S_ for k in range(T_)])
    node.sync()
    oras = [O.PosePrior(params) for _ in range(S_)]
    dead = [False] * S_
    for k, t in enumerate(ts):
        ra = node.fetch_stage2_walk(k * S_, S_)
        for s in range(S_):
            if dead[s]:
                continue
            tally["seq_steps"] += 1
            if ra[s].c.flags & 1:   # SE3D_S2FLAG track capacity exceeded: results truncated by contract, the oracle has no capacity
                tally["seq_steps_track_overflow"] += 1
                dead[s] = True
                continue
            o2 = oras[s].step(r1[k * S_ + s].published(), rig.n_cam, t, seqs[s][k].fb_delay, capacity=mh)
            try:
                compare_stage2(ra[s], o2, check_tracks=(k == T_ - 1))
            except AssertionError as e:
                msg = str(e)
                if "mm" in msg or "cov" in msg or "marg" in msg or msg == "":
                    tally["stage2_numeric_failures"] += 1   # tolerance, not an index
                    print("stage-2 numeric: trial", trial, "step", k, "seq", s, msg[:160], file=sys.stderr, flush=True)
                else:
                    # An index differs.  The one case analysed so far (seed 4, trial 2): a person whose eye and ear were
                    # triangulated 400 m away (no hip pair -> no root-distance filter) and hang on the body by the ear-shoulder bone: the LM
                    # problem is almost flat along a ray, the trio LM (sums in another order than the per-step kernel / the oracle) takes a
                    # different accept / reject path (14 instead of 9 iterations), the track state differs by centimetres and ten steps later
                    # one association differs.  SE3D_WALK_LM=solo (the per-step device code) reproduces the oracle on that sequence.
                    print("stage-2 index mismatch: trial", trial, "step", k, "seq", s, msg[:120], file=sys.stderr, flush=True)
                    tally["stage2_index_divergences"] += 1
                    dead[s] = True
    node.close()
    tally["trials"] += 1
tally["seconds"] = round(time.time() - t0, 1)
tally["seed"] = seed
tally["tolerances"] = {"POS_TOL_M": POS_TOL_M, "SCORE_RTOL": SCORE_RTOL, "COV_REL_FRO": COV_REL_FRO}
print(json.dumps(tally))
