#!/usr/bin/env python3
"""Regression vectors of the ORACLE ITSELF: the outputs of oracle/libse3d_oracle.so on two fixed scenes (stage 1 per frame, the pose_prior
callback over the sequence), as hashes plus a few plain numbers.  They pin the restatement to its own current behaviour — not to the
reference (rows T / U / R stay "parity unpinned"): a change of the oracle's arithmetic (such as round 3's JacobiSVD
square-input fix) must show up in a CPU test and be re-generated on purpose.   python tests/golden/make_oracle_regression_golden.py"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from smartedgesensor3dhumanpose_amd import person_msgs as pm, rig as R, scene as S   # noqa: E402
from smartedgesensor3dhumanpose_amd.skeleton_3d import rig_tensors                  # noqa: E402
from oracle import oracle_py as O                                                   # noqa: E402

SCENES = [dict(name="extr16_clean", rig="extr16", cfg=dict(n_persons=6, n_frames=12, seed=3)),
          dict(name="extr16_rough", rig="extr16", cfg=dict(n_persons=7, n_frames=12, seed=8, p_miss=0.15, outlier_frac=0.03, low_score_frac=0.2, pixel_noise=3.0, spacing=0.7))]


def run(scene):
    rig = R.extr_rig_16cam()
    rig.set_tensors(*rig_tensors(rig))
    params = pm.Params(pm.POSE_SIMPLE, 0, 0.045)
    frames, _ = S.generate_sequence(rig, S.SceneConfig(**scene["cfg"]))
    pp = O.PosePrior(params)
    out = {"name": scene["name"], "cfg": scene["cfg"], "frames": []}
    for fr in frames:
        o = O.triangulate(rig, params, fr, capacity=64, hyp_capacity=64)
        f2 = pp.step(o.published(), rig.n_cam, fr.stamp, fr.fb_delay, capacity=64)
        n = f2.n_persons
        out["frames"].append({
            "n_hyp": int(o.n_hyp), "n_persons": int(o.n_persons), "flags": int(o.flags),
            "assoc_sha": hashlib.sha256(np.ascontiguousarray(o.assoc()).tobytes()).hexdigest()[:16],
            "persons_sha": hashlib.sha256(o.published().tobytes()).hexdigest()[:16],
            "first_joint": [float(v) for v in o.published()[0]["keypoints"]["joint"][0]] if o.n_persons else None,
            "fused_n": int(n), "n_tracks": int(f2.c.n_tracks),
            "fused_sha": hashlib.sha256(f2.fused[:n].tobytes() + f2.fused_pred[:n].tobytes()).hexdigest()[:16],
            "lm_iterations": [int(v) for v in f2.lm_iterations[:n]]})
    return out


if __name__ == "__main__":
    res = {"generator": "tests/golden/make_oracle_regression_golden.py", "scenes": [run(s) for s in SCENES]}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_regression.json")
    json.dump(res, open(path, "w"), indent=1)
    print("wrote", path, sum(len(s["frames"]) for s in res["scenes"]), "frames")
