"""Camera rigs: the reference's real 16-camera rig (fixture data) and seeded elliptical synthetic rigs."""
import json
import os

import numpy as np

_DATA = os.path.join(os.path.dirname(__file__), "data")


class CameraRig:
    """Static per-rig tensors of skeleton_3d main() (skeleton_3d_triang_mult_node.cpp:1184-1214).

    T      [n,4,4] float64 base->camera transforms (what tf lookupTransform(cam, base) returns, S3D:166)
    K      [n,4]   float32 fx, fy, cx, cy (distortion ignored like the reference, S3D:199-228)
    Rt, F  float32 tensors; filled by whoever owns the numerics (product host code or the oracle) so that
           both parity paths consume bit-identical inputs.
    """

    def __init__(self, T, K, width=640, height=480, name="rig"):
        self.T = np.ascontiguousarray(T, np.float64)
        self.n_cam = self.T.shape[0]
        self.K = np.ascontiguousarray(K, np.float32).reshape(self.n_cam, 4)
        self.width, self.height = width, height
        self.name = name
        self.Rt = None
        self.F = None

    @property
    def n_pairs(self):
        return self.n_cam * (self.n_cam - 1) // 2

    def centres(self):
        return np.stack([-(t[:3, :3].T @ t[:3, 3]) for t in self.T])

    def set_tensors(self, Rt, F):
        self.Rt = np.ascontiguousarray(Rt, np.float32).reshape(self.n_cam, 12)
        self.F = np.ascontiguousarray(F, np.float32).reshape(self.n_pairs, 9)
        return self


def pair_index(i, j, n):
    """get_fundamental_idx (skeleton_3d_triang_mult_node.cpp:242-253), closed form."""
    assert 0 <= i < j < n
    return i * n - i * (i + 1) // 2 + (j - i - 1)


def extr_rig_16cam():
    """The 16-camera rig of pose_prior/launch/cameras_extrinsics.launch (data extracted by tests/golden/make_extr_rig.py)."""
    d = json.load(open(os.path.join(_DATA, "extr_rig_16cam.json")))
    T = np.array(d["T_base_to_cam"], np.float64)
    k = d["intrinsics_assumed"]
    K = np.tile(np.array([k["fx"], k["fy"], k["cx"], k["cy"]], np.float32), (T.shape[0], 1))
    return CameraRig(T, K, k["width"], k["height"], name="extr16")


def look_at(centre, target, up=(0.0, 0.0, 1.0)):
    """base->camera transform of a pinhole camera at `centre` looking at `target` (optical frame: x right, y down, z fwd)."""
    c = np.asarray(centre, np.float64)
    z = np.asarray(target, np.float64) - c
    z /= np.linalg.norm(z)
    x = np.cross(z, np.asarray(up, np.float64))
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z])
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = -R @ c
    return T


def elliptical_rig(n_cam, seed=0, a=12.0, b=8.0, z=2.5, jitter=0.3, fx=600.0, width=640, height=480):
    """N cameras evenly spaced on an ellipse (semi-axes a x b, height z) +- U(-jitter, jitter), looking at (0,0,1)."""
    rng = np.random.default_rng(1000 + seed)
    T = []
    for i in range(n_cam):
        th = 2 * np.pi * i / n_cam
        c = np.array([a * np.cos(th), b * np.sin(th), z]) + rng.uniform(-jitter, jitter, 3)
        T.append(look_at(c, (0.0, 0.0, 1.0)))
    K = np.tile(np.array([fx, fx, width / 2.0, height / 2.0], np.float32), (n_cam, 1))
    return CameraRig(np.stack(T), K, width, height, name="ellipse%d" % n_cam)
