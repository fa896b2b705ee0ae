#!/usr/bin/env python3
"""Round 6 study (GPU box): on which stage-2 inputs do the library's LM and the oracle's stop >= 1 mm apart on a PUBLISHED person, so that the
solver-independent criterion of tests/parity.py (compare_stage2_objectives: the library's final objective must not be the worse one) has something to
judge?  Clean stage-1 persons of the 16-camera rig, one joint of one person replaced by a GARBAGE joint: `dist` metres along the ray from a camera
through the true joint, covariance sigma^2 along that ray and (2 cm)^2 across — what two nearly parallel rays triangulate to (hard part 4).
Prints per (dist, sigma): persons compared, persons with joints >= 1 mm apart, of them with the worse objective on the library's side."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle_py as O                                            # noqa: E402
from smartedgesensor3dhumanpose_amd import person_msgs as pm, rig as R, scene as S   # noqa: E402
from smartedgesensor3dhumanpose_amd.pose_prior import PosePrior              # noqa: E402
from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D, rig_tensors   # noqa: E402
from parity import compare_stage2_objectives                                 # noqa: E402
import branch_scenes                                                         # noqa: E402


def main():
    rig = R.extr_rig_16cam()
    rig.set_tensors(*rig_tensors(rig))
    params = pm.Params(pm.POSE_SIMPLE, 0, 0.045)
    frames, _ = S.generate_sequence(rig, S.SceneConfig(n_persons=4, n_frames=16, seed=3))
    s1 = [O.triangulate(rig, params, fr, capacity=32, hyp_capacity=32).published() for fr in frames]
    for dist in (0.0, 0.3):
        for sigma in (3e2, 1e3, 3e3, 1e4, 1e5):
            node = Skeleton3D(rig, max_epi_dist=0.045, max_frames=1, max_det_per_cam=8, max_hyp=16, device=0)
            pp, ora = PosePrior(node, 0), O.PosePrior(params)
            tot = dict(persons=0, disagree=0, worse=0, not_comparable=0, worst=0.0, max_mm=0.0)
            for k, fr in enumerate(frames):
                dets = branch_scenes.garbage_joint(rig, s1[k], person=0, joint=9, dist=dist, sigma=sigma)
                g = pp.skeleton_callback(dets, fr.stamp, fr.fb_delay)
                o = ora.step(dets, rig.n_cam, fr.stamp, fr.fb_delay, capacity=16)
                oc = compare_stage2_objectives(g, o)
                if not oc["comparable"]:
                    tot["not_comparable"] += 1
                    continue
                tot["persons"] += oc["persons"]; tot["disagree"] += oc["disagree"]; tot["worse"] += oc["worse"]; tot["worst"] = max(tot["worst"], oc["worst_excess"])
                n = g.n_persons
                if n and oc["disagree"]:
                    dd = np.linalg.norm(g.fused[:n]["keypoints"]["joint"] - o.fused[:n]["keypoints"]["joint"], axis=-1)
                    vv = o.fused[:n]["keypoints"]["score"] > 0
                    for p in range(n):
                        if (dd[p] * vv[p]).max() >= 1e-3:
                            print("   step %d person %d: max joint diff %.3f mm at joint %d, objective lib %.9g oracle %.9g, iterations lib %d oracle %d, ids %d %d"
                                  % (k, p, (dd[p] * vv[p]).max() * 1e3, int(np.argmax(dd[p] * vv[p])), g.lm_objective[p], o.lm_objective[p], g.lm_iterations[p], o.lm_iterations[p],
                                     g.fused["id"][p], o.fused["id"][p]))
                if n:
                    v = o.fused[:n]["keypoints"]["score"] > 0
                    d = np.linalg.norm(g.fused[:n]["keypoints"]["joint"] - o.fused[:n]["keypoints"]["joint"], axis=-1)
                    tot["max_mm"] = max(tot["max_mm"], float(d[v].max() * 1e3))
            node.close()
            print("dist %6.1f m sigma %6.1f m: %s" % (dist, sigma, tot), flush=True)


if __name__ == "__main__":
    main()
