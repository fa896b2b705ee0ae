"""CPU tests of the rviz marker builders (host/se3d_markers.hpp): the MarkerArray outputs of the two nodes
(skeleton_3d_triang_mult_node.cpp:689-716, 858-921, 967-971; pose_prior_mult_node.cpp:273-375, 598-627, 191-209), built on the host from the
persons the accelerated path returns.  The driver places joint k at x = k, so every printed point names its joint."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE_LIST, SPHERE_LIST, ADD, DELETE = 5, 7, 0, 2


@pytest.fixture(scope="module")
def driver():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host"), "se3d_markers_test"], stdout=subprocess.DEVNULL)
    return os.path.join(ROOT, "host", "se3d_markers_test")


def run(driver, lines):
    out = subprocess.run([driver], input="\n".join(lines) + "\n", capture_output=True, text=True, check=True).stdout
    groups, cur = [], []
    for ln in out.strip().split("\n"):
        if ln == "--":
            groups.append(cur)
            cur = []
            continue
        head, pts, tail = ln.split("|")
        ns, mid, typ, act, life = head.split()
        cur.append(dict(ns=ns, id=int(mid), type=int(typ), action=int(act), lifetime=float(life), points=[int(x) for x in pts.split()],
                        n_colors=int(tail.split()[0]), first_color=[float(x) for x in tail.split()[-3:]]))
    return groups


def segments(marker):
    p = marker["points"]
    assert len(p) % 2 == 0 and marker["n_colors"] == len(p)
    return list(zip(p[0::2], p[1::2]))


def test_stage1_markers_follow_the_parent_table(driver):
    const = json.load(open(os.path.join(ROOT, "tests", "golden", "pose_constants.json")))   # tables parsed from the reference's sources
    for method, mask in (("simple", 0x7ffff), ("h36m", 0x181f7f)):
        kp2fu = const[method]["kp2fusion"]
        parent = [int(v) for v in const[method]["kpParent"]]
        mask_all = sum(1 << f for f in kp2fu)
        (g,) = run(driver, ["s1 %s %x" % (method, mask_all)])
        skel, joints = g
        assert (skel["ns"], skel["type"], skel["lifetime"]) == ("joints", LINE_LIST, 2.0) and (joints["ns"], joints["type"]) == ("joint_spheres", SPHERE_LIST)
        assert joints["points"] == kp2fu                                   # one sphere per keypoint in input-keypoint order (S3D:885)
        expect = [(kp2fu[parent[k]] if parent[k] >= 0 else kp2fu[k], kp2fu[k]) for k in range(17)]
        assert segments(skel) == expect                                    # parent -> joint, the root as a zero-length segment (S3D:901-918)
        assert skel["first_color"] == [1.0, 0.0, 0.0]                      # nose: (255, 0, 0)
    # a joint whose parent is missing is drawn as a zero-length segment (S3D:902-905): simple without the nose
    (g,) = run(driver, ["s1 simple %x" % (0x7ffff & ~1)])
    assert (16, 16) in segments(g[0]) and (5, 5) in segments(g[0]) and (16, 18) in segments(g[0])


def test_fused_markers_and_fallback_neighbours(driver):
    g = run(driver, ["s2 7 10.0 1fffff", "s2 7 10.5 %x" % (0x1fffff & ~(1 << 8)), "s2 7 11.0 %x" % (0x1fffff & ~((1 << 8) | (1 << 1)))])
    joints, skel = g[0]
    assert (joints["ns"], joints["id"], joints["type"]) == ("joints_fused", 7, SPHERE_LIST) and (skel["ns"], skel["id"], skel["type"]) == ("skeleton_fused", 7, LINE_LIST)
    assert joints["points"] == list(range(21)) and skel["lifetime"] == 0.0
    s = segments(skel)
    assert len(s) == 22                                                    # one per joint, two for the belly (PP:318-336)
    for seg in [(0, 0), (0, 1), (1, 2), (2, 3), (3, 4), (1, 5), (1, 8), (8, 9), (9, 10), (10, 11), (8, 12), (0, 15), (15, 17), (16, 18), (0, 19), (1, 20), (8, 20)]:
        assert seg in s
    s = segments(g[1][1])                                                  # no mid hip: hips hang on the neck, the belly's second segment collapses
    assert (1, 9) in s and (1, 12) in s and (20, 20) in s and (1, 20) in s
    s = segments(g[2][1])                                                  # neither mid hip nor neck: shoulders on the nose, hips on their shoulders
    assert (0, 2) in s and (0, 5) in s and (2, 9) in s and (5, 12) in s


def test_delete_markers_after_the_track_timeout(driver):
    g = run(driver, ["s2 3 5.0 1fffff", "s2 - 5.9 0", "s2 - 6.05 0", "s2 - 6.2 0"])
    assert len(g[0]) == 2 and g[1] == []
    assert [(m["ns"], m["id"], m["action"]) for m in g[2]] == [("skeleton_fused", 3, DELETE), ("joints_fused", 3, DELETE)]   # unseen for > 1.0 s (PP:62, 191-209)
    assert g[3] == []                                                      # once


def test_vis_cov_ellipsoid_markers(driver):
    """`vis_cov` (S3D:279-309, 887-898; PP:237-254, 795-807): a SPHERE marker per joint whose pose / scale are the principal axes of the joint's
    covariance: R(q) diag((scale / (2 * 2.7955))^2) R(q)^T must give the covariance back, R a proper rotation, scales ascending (Eigen's
    SelfAdjointEigenSolver order), eyes / ears / head / belly (fusion part >= 15) not drawn, the mid hip's x 100^2 undone for the fused marker."""
    import numpy as np
    rng = np.random.default_rng(3)

    def quat_to_R(q):   # x, y, z, w
        x, y, z, w = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])

    def call(stage, part, C):
        c6 = [C[0, 0], C[0, 1], C[0, 2], C[1, 1], C[1, 2], C[2, 2]]
        out = subprocess.run([driver], input="cov %s %d %s\n" % (stage, part, " ".join("%.17g" % v for v in c6)), capture_output=True, text=True, check=True).stdout
        return [ln for ln in out.strip().split("\n") if ln != "--"]

    for trial in range(20):
        A = rng.normal(size=(3, 3)) * rng.uniform(0.001, 0.3)
        C = A @ A.T + 1e-8 * np.eye(3)
        if trial == 0:
            C = np.diag([4e-4, 1e-4, 9e-4])            # axis-aligned: eigenvalues out of order
        if trial == 1:
            C = 2.5e-3 * np.eye(3)                      # isotropic: any rotation is right
        for stage, part in (("s1", 4), ("s2", 8), ("s2", 3)):   # RWrist (simple keypoint 10), MidHip (rescale), RElbow
            lines = call(stage, part, C)
            head, pos, quat, scale = [x.split() for x in lines[0].split("|")]
            ns, mid, typ, life, alpha = head
            assert typ == "2" and float(alpha) == 0.5
            if stage == "s1":
                assert ns == "joint_cov_3d" and int(mid) == 21 * 1 + 10 and float(life) == 5.0
            else:
                assert ns == "joint_cov_fused" and int(mid) == 21 * 1 + part and float(life) == 1.0
            assert [float(v) for v in pos] == [1.0, 2.0, 3.0]
            q = np.array([float(v) for v in quat])
            s = np.array([float(v) for v in scale])
            R = quat_to_R(q)
            assert abs(np.linalg.norm(q) - 1) < 1e-12 and abs(np.linalg.det(R) - 1) < 1e-9
            assert s[0] <= s[1] * (1 + 1e-12) and s[1] <= s[2] * (1 + 1e-12)
            want = C / 1e4 if (stage, part) == ("s2", 8) else C
            got = R @ np.diag((s / (2 * 2.7955)) ** 2) @ R.T
            assert np.allclose(got, want, rtol=1e-9, atol=1e-12 * np.abs(want).max())
            if stage == "s1":
                ev = np.linalg.eigvalsh(C)
                rng_mm = [float(v) for v in lines[1].split()[1:]]
                assert np.allclose(rng_mm[:3], np.sqrt(ev) * 1e3, rtol=1e-6) and np.allclose(rng_mm[3:], np.sqrt(ev) * 1e3, rtol=1e-6)
    assert call("s1", 16, np.eye(3) * 1e-3) == []     # LEye (fusion part 16): not drawn
    assert call("s2", 17, np.eye(3) * 1e-3) == []
