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
