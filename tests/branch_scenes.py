"""Scenes that FORCE the data-dependent branches of the reference: every builder returns inputs on which the oracle's
branch counter (oracle_py.stats()) for that branch is > 0.  Used by tests/test_oracle_branches.py (CPU: the branch fires in the oracle)
and tests/test_gpu_branches.py (GPU: the HIP path takes the same branch and agrees with the oracle)."""
import numpy as np

from smartedgesensor3dhumanpose_amd import scene as S

SEED = 11   # three well-covered spots on the 16-camera rig


def _spots(rig, n):
    return S.generate_sequence(rig, S.SceneConfig(n_persons=n, n_frames=1, seed=SEED))[1]["positions"]


def duplicate_merge(rig, n_frames=3, offset=0.12):
    """Two people 12 cm apart with the same pose: two hypotheses survive the sweep, their triangulated skeletons are < 0.20 m apart
    -> merge_persons (S3D:984-996, 410-423)."""
    pos = _spots(rig, 3)
    p2 = np.vstack([pos, pos[0] + [offset, 0.02]])
    cfg = S.SceneConfig(n_persons=4, n_frames=n_frames, seed=SEED, positions=p2, yaws=[0.3, 1.0, 2.0, 0.3], velocities=np.zeros((4, 2)))
    return S.generate_sequence(rig, cfg)[0]


def root_distance_reset(rig, n_frames=2):
    """Left wrist of person 0 consistently 2.5 m away from the body in every view: it triangulates fine and is then reset because it is
    > 2.0 m from the root (S3D:937-953)."""
    jo = np.zeros((3, 17, 3))
    jo[0, 9] = [0.0, 2.5, 0.0]
    return S.generate_sequence(rig, S.SceneConfig(n_persons=3, n_frames=n_frames, seed=SEED, joint_offset=jo))[0]


def feet_height_reject(rig, n_frames=2):
    """Person 1 floats 0.6 m above the ground: |feet height| > 0.5 m rejects the person (S3D:955-966)."""
    jo = np.zeros((3, 17, 3))
    jo[1, :, 2] = 0.6
    return S.generate_sequence(rig, S.SceneConfig(n_persons=3, n_frames=n_frames, seed=SEED, joint_offset=jo))[0]


def outliers(rig, n_frames=2, p_miss=0.0):
    """Gross 2-D errors on 8 % of the joints: 3-view epipolar rejection, leave-one-out rejection, score down-weighting, hypotheses spawned
    from infeasible assignments (S3D:748-844, 640-648).  p_miss = 0.6 leaves many joints with exactly 3 views."""
    return S.generate_sequence(rig, S.SceneConfig(n_persons=3, n_frames=n_frames, seed=SEED, outlier_frac=0.08, p_miss=p_miss))[0]


def non_spd_cov2d(rig, n_frames=2):
    """Every third detection carries an indefinite 2x2 covariance (xx 4, xy 3, yy 1) on its left shoulder (S3D:473-475 takes the square
    root of a negative number): defined policy = NaN 3-D covariance for that joint."""
    frames = S.generate_sequence(rig, S.SceneConfig(n_persons=3, n_frames=n_frames, seed=SEED))[0]
    for fr in frames:
        c = fr.persons["keypoints"]["cov"]
        c[::3, 5, 1] = 3.0
        c[::3, 5, 2] = 1.0
    return frames


def _pick(dets, gt_joints):
    g = gt_joints.mean(0)
    d = [np.linalg.norm(x["keypoints"]["joint"][x["keypoints"]["score"] > 0].mean(0) - g) for x in dets]
    return dets[int(np.argmin(d))]


def converging_tracks(rig, stage1, n_frames=24):
    """PersonCov lists for pose_prior: person 0, a clone of it 0.6 m away for 14 frames (both tracks get published) that then closes in to
    5 cm, and person 1.  The clone's track is merged into person 0's once they are < 0.20 m apart and the id already written into the
    published lists is rewritten (PP:870-903).  stage1(frame) -> published person3d array of one frame (oracle or GPU stage 1)."""
    frames, gt = S.generate_sequence(rig, S.SceneConfig(n_persons=2, n_frames=n_frames, seed=SEED, velocities=np.zeros((2, 2))))
    out = []
    for k, fr in enumerate(frames):
        dets = stage1(fr)
        a, c = _pick(dets, gt["joints"][k, 0]).copy(), _pick(dets, gt["joints"][k, 1]).copy()
        b = a.copy()
        off = 0.6 if k < 14 else max(0.05, 0.6 - 0.15 * (k - 13))
        b["keypoints"]["joint"][..., 0] += np.where(b["keypoints"]["score"] > 0, off, 0.0)
        out.append((fr, np.stack([a, b, c])))
    return out


def non_spd_cov3d(dets):
    """Copy of a PersonCov list whose first person has an indefinite 3x3 covariance on its left elbow (|xy| > sqrt(xx yy)): the whitening
    of that unary factor fails -> 'optimisation failed', the measurements are published with the default covariance (PP:751-758, 788-793)."""
    d = dets.copy()
    d[0]["keypoints"]["cov"][6] = [0.01, 0.02, 0.0, 0.01, 0.0, 0.01]
    return d


def garbage_joint(rig, dets, person=0, joint=9, dist=30.0, sigma=30.0, cam=0):
    """Stage-2 input with a GARBAGE joint (round 6): fusion joint `joint` of stage-1 person `person` moved `dist` metres along the ray from camera
    `cam` through it, its covariance sigma^2 along that ray and (2 cm)^2 across — what two nearly parallel rays triangulate to on a person whose
    root survives the plausibility filters (S3D:937-953 checks the distance to the root only when the root exists; SURVEY 7 hard part 4).  The
    person's LM problem (PP:690-746) is flat along the ray: algebraically equivalent solvers stop at different points, which is what the
    solver-independent objective criterion of tests/parity.py is for.  Returns a copy of `dets`."""
    out = np.array(dets, copy=True)
    if len(out) <= person or out[person]["keypoints"]["score"][joint] <= 0:
        return out
    c = rig.centres()[cam]
    x = out[person]["keypoints"]["joint"][joint]
    u = (x - c) / np.linalg.norm(x - c)
    out[person]["keypoints"]["joint"][joint] = x + dist * u
    a, b = np.cross(u, [0.0, 0.0, 1.0]), None
    a /= np.linalg.norm(a)
    b = np.cross(u, a)
    C3 = sigma ** 2 * np.outer(u, u) + 0.02 ** 2 * (np.outer(a, a) + np.outer(b, b))
    out[person]["keypoints"]["cov"][joint] = [C3[0, 0], C3[0, 1], C3[0, 2], C3[1, 1], C3[1, 2], C3[2, 2]]
    return out
