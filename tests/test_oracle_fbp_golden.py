"""Row K (constants) pinned to the reference: tests/golden/fusion_body_parts.json holds the FUSION_BODY_PARTS joint indices and the
vel_sigmas table read from the reference's own fusion_body_parts.h compiled in place (tests/golden/make_fbp_golden.py, oracle/_ref).
CPU: the oracle's restated constants equal the golden values (and, in the build container, the freshly compiled reference header).
GPU: so do the constants compiled into the HIP kernels."""
import json
import os

import pytest

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "fusion_body_parts.json")))
NAMES = ["Nose", "Neck", "RShoulder", "RElbow", "RWrist", "LShoulder", "LElbow", "LWrist", "MidHip", "RHip", "RKnee", "RAnkle", "LHip", "LKnee",
         "LAnkle", "REye", "LEye", "REar", "LEar", "Head", "Belly", "NUM_KEYPOINTS"]


def test_oracle_fusion_constants_equal_the_reference_header(oracle):
    idx, vel = oracle.fusion_constants()
    assert idx == [GOLD["indices"][n] for n in NAMES]
    assert vel == GOLD["vel_sigmas"]
    if oracle.ref_fbp_available():   # build container: the golden file still matches the reference tree
        ridx, rvel = oracle.ref_fusion_constants()
        assert ridx == idx and rvel == vel


def test_person_msgs_fusion_tables_use_the_reference_indices():
    from smartedgesensor3dhumanpose_amd import person_msgs as pm
    g = GOLD["indices"]
    assert pm.NUM_FUSION_KEYPOINTS == g["NUM_KEYPOINTS"]
    # the simple keypoint set maps onto the reference's named joints (S3D:139-142: Nose, LEye, REye, LEar, REar, LShoulder, RShoulder, ...)
    want = [g[n] for n in ("Nose", "LEye", "REye", "LEar", "REar", "LShoulder", "RShoulder", "LElbow", "RElbow", "LWrist", "RWrist", "LHip", "RHip",
                           "LKnee", "RKnee", "LAnkle", "RAnkle")]
    assert list(pm.KP2FUSION_SIMPLE) == want


@pytest.mark.gpu
def test_kernel_fusion_constants_equal_the_reference_header(extr_rig):
    from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
    node = Skeleton3D(extr_rig, max_frames=1, max_det_per_cam=8, max_hyp=8)
    idx, vel = node.fusion_constants()
    node.close()
    want = [GOLD["indices"][n] for n in NAMES]
    assert all(i == -1 or i == w for i, w in zip(idx, want)) and idx[21] == want[21] and sum(i >= 0 for i in idx) >= 10
    assert vel == GOLD["vel_sigmas"]


def test_demo_configuration_is_the_reference_launch_file(demo_params, extr_rig):
    """tests/golden/demo_params.json: arguments of the reference's pose_triangulate_demo.launch (make_demo_params_golden.py).  The
    parameters every parity test and bench.py run with, and the 16-camera rig, are that configuration."""
    from smartedgesensor3dhumanpose_amd import person_msgs as pm
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "demo_params.json")))
    assert demo_params.max_epi_dist == gold["max_epi_dist"] and bool(demo_params.norm_height) == gold["norm_height"]
    assert gold["pose_methods"] == ["simple"] and demo_params.pose_method == pm.POSE_SIMPLE
    assert extr_rig.n_cam == gold["n_cameras"]


def test_oracle_node_constants_and_pose_tables_equal_the_reference_sources(oracle):
    """tests/golden/pose_constants.json: the named constants (g_*) and the EdgeTPU_BodyParts_* / g_kp2kpFusion_idx_* tables of the reference's
    node sources, by name (make_constants_golden.py).  The oracle's restated values — the ones every kernel is compared against — are those;
    so are the Python mirror's keypoint maps."""
    import numpy as np
    from smartedgesensor3dhumanpose_amd import person_msgs as pm
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pose_constants.json")))
    vals = oracle.node_constants()
    for (src, name), v in zip(oracle.NODE_CONSTANT_NAMES, vals):
        assert np.float32(v) == np.float32(gold[src][name]) if name in ("g_min_score", "g_triangulation_threshold") else v == gold[src][name], (src, name, v)
    assert gold["skeleton_3d"]["g_merge_dist_thresh"] == gold["pose_prior"]["g_merge_dist_thresh"]
    for pose, key in ((pm.POSE_SIMPLE, "simple"), (pm.POSE_H36M, "h36m")):
        t = oracle.pose_tables(pose)
        for name in ("kpParent", "limbLength", "limbLSigma", "kp2fusion"):
            assert [float(x) for x in t[name]] == [float(x) for x in gold[key][name]], (key, name)
    t = oracle.pose_tables(pm.POSE_SIMPLE)
    assert t["shoulderDist"] == gold["simple"]["shoulderDist"] and t["shoulderSigma"] == gold["simple"]["shoulderSigma"]
    assert list(pm.KP2FUSION_SIMPLE) == gold["simple"]["kp2fusion"]
    if hasattr(pm, "KP2FUSION_H36M"):
        assert list(pm.KP2FUSION_H36M) == gold["h36m"]["kp2fusion"]



def test_oracle_bone_tables_equal_the_reference_source(oracle):
    """The 2 x 21 range factors of addBinaryFactors (PP:384-481) — joint pair, length, sigma coefficient, measured / not-measured conditions —
    parsed out of the reference source in order (tests/golden/pose_constants.json["bones"]) against the oracle's tables."""
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pose_constants.json")))
    belly = GOLD["indices"]["Belly"]
    for norm, key in ((False, "absolute"), (True, "normalized")):
        tab = oracle.bone_table(norm)
        assert len(gold["bones"][key]) == 21
        for (k1, k2, length, sigma, not_belly), (g1, g2, gl, gs, need, deny) in zip(tab, gold["bones"][key]):
            assert (k1, k2, length, sigma) == (g1, g2, gl, gs)
            assert sorted(need) == sorted([k1, k2]) and deny == ([belly] if not_belly else [])
