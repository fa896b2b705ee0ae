#!/usr/bin/env python3
"""Generates smartedgesensor3dhumanpose_amd/data/extr_rig_16cam.json from the reference's launch file.

Source of the numbers: /root/reference/pose_prior/launch/cameras_extrinsics.launch:2-18 (static_transform_publisher
args "x y z qx qy qz qw parent child"; cam_1 is given in `base`, cams 2..16 relative to cam_1_color_optical_frame; tf2
convention: pose of child in parent).  The skeleton_3d node uses lookupTransform(cam, base) = base->camera
(skeleton_3d_triang_mult_node.cpp:166), i.e. the inverse of the composed pose.  Camera order = the `cameras` list of
pose_triangulate_demo.launch:5 (cam_1 .. cam_16).  Only configuration DATA is extracted; run in the build container.
"""
import json, re, sys, os
import numpy as np

REF = "/root/reference/pose_prior/launch/cameras_extrinsics.launch"
OUT = os.path.join(os.path.dirname(__file__), "..", "..", "smartedgesensor3dhumanpose_amd", "data", "extr_rig_16cam.json")


def quat_to_rot(qx, qy, qz, qw):
    n = np.sqrt(qx * qx + qy * qy + qz * qz + qw * qw)
    qx, qy, qz, qw = qx / n, qy / n, qz / n, qw / n
    return np.array([
        [1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
        [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
        [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])


def main():
    txt = open(REF).read()
    poses = {}
    for m in re.finditer(r'args="([^"]+)"', txt):
        a = m.group(1).split()
        vals = list(map(float, a[:7]))
        parent, child = a[7].lstrip("/"), a[8].lstrip("/")
        T = np.eye(4)
        T[:3, :3] = quat_to_rot(*vals[3:7])
        T[:3, 3] = vals[:3]
        poses[child] = (parent, T)
    T_base_cam1 = poses["cam_1_color_optical_frame"][1]
    out = []
    for i in range(1, 17):
        parent, T = poses["cam_%d_color_optical_frame" % i]
        T_base_cam = T if parent == "base" else T_base_cam1 @ T
        out.append(np.linalg.inv(T_base_cam).tolist())  # base -> camera
    json.dump({"source": "pose_prior/launch/cameras_extrinsics.launch:2-18", "order": ["cam_%d" % i for i in range(1, 17)],
               "T_base_to_cam": out,
               "intrinsics_assumed": {"fx": 600.0, "fy": 600.0, "cx": 320.0, "cy": 240.0, "width": 640, "height": 480}},
              open(OUT, "w"), indent=1)
    C = np.array([np.linalg.inv(np.array(t))[:3, 3] for t in out])
    print("camera centres span", C.min(0), C.max(0))


if __name__ == "__main__":
    main()
