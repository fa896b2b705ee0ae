"""GPU (-m gpu): config 2 of the benchmark — se3d_triangulate_assoc (rows N, T, P, O, U with the association given) through
the C-ABI against the oracle's se3d_oracle_triangulate_assoc on the same inputs: joints < 1 mm, scores bit-equal, UT covariance within
the stage-1 tolerance, the same joints (un)triangulated."""
import os
import sys

import numpy as np
import pytest

from smartedgesensor3dhumanpose_amd import person_msgs as pm
from smartedgesensor3dhumanpose_amd import rig as R
from smartedgesensor3dhumanpose_amd import scene as S
from smartedgesensor3dhumanpose_amd.skeleton_3d import rig_tensors

sys.path.insert(0, os.path.dirname(__file__))
from parity import COV_REL_FRO, POS_TOL_M, SCORE_RTOL  # noqa: E402

pytestmark = pytest.mark.gpu


def _node(rig, **kw):
    from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
    return Skeleton3D(rig, max_epi_dist=0.045, **kw)


def _compare(got, ref):
    g, o = got["keypoints"], ref["keypoints"]
    vg, vo = g["score"] > 0, o["score"] > 0
    assert np.array_equal(vg, vo), "set of triangulated joints differs"
    assert got[..., None]["id"].max() == 0
    if not vo.any():
        return 0.0, 0.0
    d = np.linalg.norm(g["joint"] - o["joint"], axis=-1)[vo]
    assert d.max() < POS_TOL_M, "max joint error %.4f mm" % (d.max() * 1e3)
    assert SCORE_RTOL == 0.0 and np.array_equal(g["score"][vo], o["score"][vo])
    assert (g["joint"][~vo] == 0).all() and (g["cov"][~vo] == 0).all()   # default-constructed slots
    cg, co = g["cov"][vo], o["cov"][vo]
    w = np.array([1, 2, 2, 1, 2, 1.0])
    rel = np.sqrt(((cg - co) ** 2 * w).sum(-1)) / np.sqrt((co ** 2 * w).sum(-1))
    assert rel.max() < COV_REL_FRO
    return float(d.max() * 1e3), float(rel.max())


def _assoc_tables(oracle, rig, params, frames, n_hyp):
    out = np.full((len(frames), n_hyp, rig.n_cam), -1, np.int32)
    for k, fr in enumerate(frames):
        a = oracle.triangulate(rig, params, fr, capacity=n_hyp, hyp_capacity=n_hyp).assoc()
        out[k, : len(a)] = a
    return out


@pytest.mark.parametrize("tri", ["hyp", "joint"])
@pytest.mark.parametrize("extra", [{}, {"outlier_frac": 0.06}, {"p_miss": 0.5, "low_score_frac": 0.2}])
def test_dlt_given_association_16cam(extr_rig, oracle, demo_params, extra, tri):
    frames = sum((S.generate_sequence(extr_rig, S.SceneConfig(n_persons=6, n_frames=2, seed=seed, **extra))[0] for seed in (0, 1, 2)), [])
    n_hyp = 24
    assoc = _assoc_tables(oracle, extr_rig, demo_params, frames, n_hyp)
    node = _node(extr_rig, max_frames=len(frames), max_det_per_cam=8, max_hyp=n_hyp, options=dict(tri_form="joint" if tri == "joint" else "auto"))
    got = node.triangulate_assoc(frames, assoc)
    worst = (0.0, 0.0)
    for k, fr in enumerate(frames):
        ref = oracle.triangulate_assoc(extr_rig, demo_params, fr, assoc[k])
        worst = tuple(max(a, b) for a, b in zip(worst, _compare(got[k], ref)))
    print("config 2 (%s, %s): max joint error %.5f mm, max covariance rel. error %.2e" % (extra, tri, *worst))
    # the resident form returns the same bytes
    node.stage(frames)
    node.stage_assoc(assoc)
    node.run_dlt()
    assert node.fetch_dlt().tobytes() == got.tobytes()
    node.close()


def test_dlt_noise_free_known_answer(extr_rig, oracle):
    """noise-free projections of known 3-D points come back to < 1e-4 m (no oracle involved)."""
    cfg = S.SceneConfig(n_persons=5, n_frames=2, seed=4, pixel_noise=0.0, low_score_frac=0.0, joint_jitter=0.0, shuffle=False)
    frames, gt = S.generate_sequence(extr_rig, cfg)
    assoc = np.full((2, 5, extr_rig.n_cam), -1, np.int32)
    for k in range(2):
        for c in range(extr_rig.n_cam):
            for det, p in enumerate(gt["ids"][k][c]):
                assoc[k, p, c] = det
    node = _node(extr_rig, max_frames=2, max_det_per_cam=8, max_hyp=8)
    out = node.triangulate_assoc(frames, assoc)
    for k in range(2):
        for p in range(5):
            kp = out[k, p]["keypoints"]
            assert (kp["score"] > 0).sum() == 17
            for j, fu in enumerate(pm.KP2FUSION_SIMPLE):
                assert np.linalg.norm(kp["joint"][fu] - gt["joints"][k, p, j]) < 1e-4
    node.close()


def test_dlt_large_rig_per_joint_kernel(oracle, demo_params):
    rig = R.elliptical_rig(40, seed=2)
    rig.set_tensors(*rig_tensors(rig))   # product tensors (bit-equal to the oracle's: tests/test_abi.py)
    frames = S.generate_sequence(rig, S.SceneConfig(n_persons=10, n_frames=2, seed=9, min_coverage=6))[0]
    assoc = _assoc_tables(oracle, rig, demo_params, frames, 32)
    node = _node(rig, max_frames=2, max_det_per_cam=16, max_hyp=32)
    got = node.triangulate_assoc(frames, assoc)
    for k, fr in enumerate(frames):
        _compare(got[k], oracle.triangulate_assoc(rig, demo_params, fr, assoc[k]))
    node.close()


def test_dlt_thin_hypotheses_and_errors(extr_rig, oracle, demo_params):
    from smartedgesensor3dhumanpose_amd.skeleton_3d import Se3dError
    fr = S.generate_sequence(extr_rig, S.SceneConfig(n_persons=3, n_frames=1, seed=8))[0][0]
    a = oracle.triangulate(extr_rig, demo_params, fr, capacity=32, hyp_capacity=32).assoc()
    one_view = np.full_like(a[0], -1)
    first = np.nonzero(a[0] >= 0)[0][:1]
    one_view[first] = a[0][first]
    table = np.vstack([a, one_view[None], np.full_like(a[:1], -1)])[None]
    node = _node(extr_rig, max_frames=1, max_det_per_cam=8, max_hyp=8)
    node.triangulate_persons([fr])            # leave another result in the hypothesis slots first: nothing stale may leak
    got = node.triangulate_assoc([fr], table)
    ref = oracle.triangulate_assoc(extr_rig, demo_params, fr, table[0])
    _compare(got[0], ref)
    assert not got[0, -1].tobytes().strip(b"\0") and not got[0, -2].tobytes().strip(b"\0")
    with pytest.raises(Se3dError):
        node.triangulate_assoc([fr], np.full((1, 9, extr_rig.n_cam), -1, np.int32))   # n_hyp > limits.max_hyp
    node.stage([fr])
    with pytest.raises(Se3dError):
        node.run_dlt()                                                                  # no table staged
    node.close()


def _exact_ut_cov(Rt, pts, cov2, centre):
    """The reference's unscented transform (S3D:471-523) evaluated in float64: sigma points as the reference forms them (float32 inputs,
    kappa = 0.5, offsets along the Cholesky factor of each view's 2x2 covariance), every sample triangulated by an exact SVD, deviations from
    the given (float32) centre, equal weights 1 / (2 (2n + kappa)) (2 kappa == 1)."""
    n = len(Rt)
    P = Rt.astype(np.float64).reshape(n, 3, 4)

    def tri(xy):
        rows = []
        for c in range(n):
            for r in (xy[c, 0] * P[c, 2] - P[c, 0], xy[c, 1] * P[c, 2] - P[c, 1]):
                rows.append(r / np.linalg.norm(r))
        v = np.linalg.svd(np.array(rows))[2][-1]
        return v[:3] / v[3]
    b = np.float32(np.sqrt(np.float32(2 * n) + np.float32(0.5)))
    samples = [pts[:, :2].astype(np.float64)]
    for c in range(n):
        l11 = np.sqrt(cov2[c, 0]); l21 = cov2[c, 1] / l11; l22 = np.sqrt(cov2[c, 2] - l21 * l21)      # float32 like the reference
        for dx, dy in ((-l11 * b, -l21 * b), (np.float32(0), -l22 * b), (l11 * b, l21 * b), (np.float32(0), l22 * b)):
            s = pts[:, :2].copy()
            s[c, 0] += dx; s[c, 1] += dy
            samples.append(s.astype(np.float64))
    w = 1.0 / (2.0 * (2 * n + 0.5))
    dev = np.array([tri(s) - centre.astype(np.float64) for s in samples])
    return w * dev.T @ dev


def test_ut_covariance_error_budget(extr_rig, oracle, demo_params):
    """Why the GPU-vs-oracle covariance tolerance is 5e-3 and not SURVEY 8c's 1e-3: against the EXACT (float64) evaluation of the reference's
    unscented transform the HIP kernels are within 1e-3 (they solve every sample in fp64 and round to float32 like the reference), the
    float32 oracle — i.e. the reference's own arithmetic — is the noisier side."""
    frames = sum((S.generate_sequence(extr_rig, S.SceneConfig(n_persons=6, n_frames=1, seed=seed))[0] for seed in (0, 1, 2)), [])
    assoc = _assoc_tables(oracle, extr_rig, demo_params, frames, 12)
    node = _node(extr_rig, max_frames=len(frames), max_det_per_cam=8, max_hyp=12)
    got = node.triangulate_assoc(frames, assoc)
    node.close()
    w6 = np.array([1, 2, 2, 1, 2, 1.0])
    K = extr_rig.K
    worst_gpu, worst_ora, n = 0.0, 0.0, 0
    for k, fr in enumerate(frames):
        ora = oracle.triangulate_assoc(extr_rig, demo_params, fr, assoc[k])
        for h in range(6):
            for j in (0, 5, 9, 12, 16):   # nose, shoulder, wrist, hip, ankle
                fu = pm.KP2FUSION_SIMPLE[j]
                if not ora[h]["keypoints"]["score"][fu] > 0:
                    continue
                cams, pts, cov2 = [], [], []
                for c in range(extr_rig.n_cam):
                    d = assoc[k, h, c]
                    if d < 0:
                        continue
                    kp = fr.persons[fr.det_offset[c] + d]["keypoints"][j]
                    if kp["score"] < np.float32(0.30):
                        continue
                    fx, fy, cx, cy = [np.float32(v) for v in K[c]]
                    cams.append(c)
                    pts.append([(kp["x"] - cx) / fx, (kp["y"] - cy) / fy, kp["score"]])
                    cov2.append([kp["cov"][0] / (fx * fx), kp["cov"][1] / (fx * fy), kp["cov"][2] / (fy * fy)])
                pts, cov2 = np.array(pts, np.float32), np.array(cov2, np.float32)
                for name, arr in (("gpu", got[k, h]), ("ora", ora[h])):   # each side against the exact evaluation about ITS OWN float32 centre
                    centre = arr["keypoints"]["joint"][fu].astype(np.float32)
                    Cx = _exact_ut_cov(extr_rig.Rt[cams], pts, cov2, centre)
                    exact6 = Cx[[0, 0, 0, 1, 1, 2], [0, 1, 2, 1, 2, 2]]
                    rel = np.sqrt((((arr["keypoints"]["cov"][fu] - exact6) ** 2) * w6).sum()) / np.sqrt(((exact6 ** 2) * w6).sum())
                    if name == "gpu":
                        worst_gpu = max(worst_gpu, rel)
                    else:
                        worst_ora = max(worst_ora, rel)
                n += 1
    print("UT covariance, relative Frobenius error against the exact float64 evaluation over %d joints: HIP %.2e, float32 oracle %.2e" % (n, worst_gpu, worst_ora))
    assert n >= 60
    assert worst_gpu < 1e-3, worst_gpu          # SURVEY 8c's bar, met against the exact arithmetic
    assert worst_gpu <= worst_ora               # the oracle's float32 SVD is the noisier side
