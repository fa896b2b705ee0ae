"""GPU (-m gpu): config 2 of the benchmark — se3d_triangulate_assoc (rows N, T, P, O, U with the association given) through
the C-ABI against the oracle's se3d_oracle_triangulate_assoc on the same inputs: joints < 1 mm, scores rtol 2e-5, UT covariance within
the stage-1 tolerance, the same joints (un)triangulated."""
import os
import sys

import numpy as np
import pytest

from smartedgesensor3dhumanpose_amd import person_msgs as pm
from smartedgesensor3dhumanpose_amd import rig as R
from smartedgesensor3dhumanpose_amd import scene as S

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
    assert np.allclose(g["score"][vo], o["score"][vo], rtol=SCORE_RTOL, atol=1e-7)
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
def test_dlt_given_association_16cam(extr_rig, oracle, demo_params, monkeypatch, extra, tri):
    if tri == "joint":
        monkeypatch.setenv("SE3D_TRI", "joint")
    frames = sum((S.generate_sequence(extr_rig, S.SceneConfig(n_persons=6, n_frames=2, seed=seed, **extra))[0] for seed in (0, 1, 2)), [])
    n_hyp = 24
    assoc = _assoc_tables(oracle, extr_rig, demo_params, frames, n_hyp)
    node = _node(extr_rig, max_frames=len(frames), max_det_per_cam=8, max_hyp=n_hyp)
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
    rig.set_tensors(*oracle.rig_tensors(rig))
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
