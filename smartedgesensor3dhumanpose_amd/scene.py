"""Seeded synthetic multi-view scenes: standing 17-joint skeletons seen by a camera rig.

The demo bag poses2D_16cam.bag of the reference is an external download and not available; these scenes are the
substitute input for every config of BASELINE.json.  Output is the exact input of triangulate_persons
(skeleton_3d_triang_mult_node.cpp:525): one ragged list of Person2D per camera and frame.
"""
from dataclasses import dataclass
from typing import Optional

import numpy as np

from .person_msgs import NUM_KEYPOINTS, Frame, person2d_dt

# canonical standing skeleton in the body frame (x forward, y left, z up), COCO-17 order of EdgeTPU_BodyParts_Simple
# (skeleton_3d_triang_mult_node.cpp:81-98); limb lengths follow its limbLength table (S3D:101), hips 0.27 m apart
# (fusion_body_parts.h:30), shoulders 0.35 m (S3D:103), ankles at z = 0.08 m.
_Z_ANKLE = 0.08
_Z_KNEE = _Z_ANKLE + 0.446
_Z_HIP = _Z_KNEE + 0.45
_Z_SHO = _Z_HIP + np.sqrt(0.50 ** 2 - 0.04 ** 2)
_Z_NOSE = _Z_SHO + 0.20
CANONICAL = np.array([
    [0.08, 0.0, _Z_NOSE],            # 0 nose
    [0.06, 0.03, _Z_NOSE + 0.035],   # 1 left eye
    [0.06, -0.03, _Z_NOSE + 0.035],  # 2 right eye
    [-0.02, 0.075, _Z_NOSE + 0.02],  # 3 left ear
    [-0.02, -0.075, _Z_NOSE + 0.02], # 4 right ear
    [0.0, 0.175, _Z_SHO],            # 5 left shoulder
    [0.0, -0.175, _Z_SHO],           # 6 right shoulder
    [0.02, 0.20, _Z_SHO - 0.279],    # 7 left elbow
    [0.02, -0.20, _Z_SHO - 0.279],   # 8 right elbow
    [0.06, 0.21, _Z_SHO - 0.279 - 0.246],   # 9 left wrist
    [0.06, -0.21, _Z_SHO - 0.279 - 0.246],  # 10 right wrist
    [0.0, 0.135, _Z_HIP],            # 11 left hip
    [0.0, -0.135, _Z_HIP],           # 12 right hip
    [0.0, 0.135, _Z_KNEE],           # 13 left knee
    [0.0, -0.135, _Z_KNEE],          # 14 right knee
    [0.0, 0.135, _Z_ANKLE],          # 15 left ankle
    [0.0, -0.135, _Z_ANKLE],         # 16 right ankle
], np.float64)


@dataclass
class SceneConfig:
    n_persons: int = 6
    n_frames: int = 1
    seed: int = 0
    dt: float = 1.0 / 30.0
    pixel_noise: float = 2.0       # N(0, sigma) px on every projected joint
    low_score_frac: float = 0.05   # joints that get score ~ U(0, 0.29): exercises the 0.30 gate (S3D:321)
    p_miss: float = 0.0            # probability that a camera misses a person entirely
    outlier_frac: float = 0.0      # probability of a gross 2D error on a visible joint (exercises S3D:748-838)
    outlier_px: float = 150.0
    min_visible_joints: int = 5    # detections with fewer visible joints are not emitted at all
    spacing: float = 1.0
    min_coverage: int = 4
    max_speed: float = 1.2
    joint_jitter: float = 0.005
    t0: float = 100.0
    shuffle: bool = True           # random per-camera detection order
    # explicit scene layout for the branch tests (None: seeded random placement as above)
    positions: Optional[np.ndarray] = None     # [P,2] ground positions
    yaws: Optional[np.ndarray] = None          # [P]
    velocities: Optional[np.ndarray] = None    # [P,2] m/s
    joint_offset: Optional[np.ndarray] = None  # [P,17,3] added to the body-frame joints after rotation (e.g. a wrist 2.5 m away, a person 0.6 m above ground)


# Rough scenes: the fixed parameters of bench.py's `robust` block — the middle of the ranges the randomised parity sweep draws from
# (draw_rough_params below; tests/fuzz_scenes.py): gross 2-D outliers exercise the reference's outlier path (S3D:748-838), missing views and
# low-confidence joints the gates of S3D:321 / 725.
ROUGH = dict(p_miss=0.08, low_score_frac=0.15, outlier_frac=0.006, pixel_noise=2.0)


def rough_config(n_persons, n_frames, seed, **kw):
    """SceneConfig of a rough scene with the fixed ROUGH parameters (keyword arguments override)."""
    return SceneConfig(n_persons=n_persons, n_frames=n_frames, seed=seed, **{**ROUGH, **kw})


def draw_rough_params(rng):
    """Scene parameters of one trial of the randomised parity sweep: up to 15 % missing views, up to 30 % low-confidence joints, up to
    1.2 % gross 2-D outliers, 0.5 - 4 px noise, persons 0.5 - 1.2 m apart.  Five draws from `rng` in this order."""
    return dict(p_miss=float(rng.uniform(0, 0.15)), low_score_frac=float(rng.uniform(0, 0.3)), outlier_frac=float(rng.uniform(0, 0.012)),
                pixel_noise=float(rng.uniform(0.5, 4.0)), spacing=float(rng.uniform(0.5, 1.2)))


def _place_persons(rig, n_persons, rng, spacing, min_coverage):
    c = rig.centres()
    lo, hi = c[:, :2].min(0) - 1.0, c[:, :2].max(0) + 1.0
    gx, gy = np.meshgrid(np.arange(lo[0], hi[0], 0.5), np.arange(lo[1], hi[1], 0.5), indexing="ij")
    cand = np.stack([gx.ravel(), gy.ravel()], 1)
    cov = np.zeros(len(cand), int)
    for zz in (0.1, 1.0, 1.7):  # feet, torso, head must be in view
        pts = np.concatenate([cand, np.full((len(cand), 1), zz)], 1)
        vis = _visible(rig, pts[None])[1][:, 0, :]  # [cam, cand]
        cov = vis.sum(0) if zz == 0.1 else np.minimum(cov, vis.sum(0))
    order = np.lexsort((rng.random(len(cand)), -cov))
    chosen = []
    for idx in order:
        if cov[idx] < min_coverage:
            break
        p = cand[idx] + rng.uniform(-0.2, 0.2, 2)
        if all(np.hypot(*(p - q)) >= spacing for q in chosen):
            chosen.append(p)
            if len(chosen) == n_persons:
                break
    if len(chosen) < n_persons:
        raise RuntimeError("rig %s cannot host %d persons at coverage >= %d (got %d)" % (rig.name, n_persons, min_coverage, len(chosen)))
    return np.array(chosen)


def _visible(rig, X):
    """Project world points X [..., 3] into every camera. Returns (uv [cam, ..., 2], visible [cam, ...])."""
    R = rig.T[:, :3, :3]
    t = rig.T[:, :3, 3]
    Xc = np.einsum("cij,...j->c...i", R, X) + t.reshape((rig.n_cam,) + (1,) * (X.ndim - 1) + (3,))
    z = Xc[..., 2]
    zs = np.where(z > 1e-6, z, 1.0)
    K = rig.K.astype(np.float64).reshape((rig.n_cam,) + (1,) * (X.ndim - 1) + (4,))
    u = K[..., 0] * Xc[..., 0] / zs + K[..., 2]
    v = K[..., 1] * Xc[..., 1] / zs + K[..., 3]
    vis = (z > 0.5) & (u >= 0) & (u < rig.width) & (v >= 0) & (v < rig.height)
    return np.stack([u, v], -1), vis


def generate_sequence(rig, cfg: SceneConfig):
    """Returns (frames, gt): frames = list of person_msgs.Frame; gt = dict with ground truth for analysis."""
    rng = np.random.default_rng(cfg.seed)
    P, T, N = cfg.n_persons, cfg.n_frames, rig.n_cam
    pos = _place_persons(rig, P, rng, cfg.spacing, cfg.min_coverage) if cfg.positions is None else np.asarray(cfg.positions, np.float64).reshape(P, 2)
    yaw = rng.uniform(0, 2 * np.pi, P)
    speed = rng.uniform(0, cfg.max_speed, P)
    heading = rng.uniform(0, 2 * np.pi, P)
    vel = np.stack([speed * np.cos(heading), speed * np.sin(heading)], 1)
    if cfg.yaws is not None:
        yaw = np.asarray(cfg.yaws, np.float64).reshape(P)
    if cfg.velocities is not None:
        vel = np.asarray(cfg.velocities, np.float64).reshape(P, 2)
    cy, sy = np.cos(yaw), np.sin(yaw)
    Rz = np.zeros((P, 3, 3))
    Rz[:, 0, 0], Rz[:, 0, 1], Rz[:, 1, 0], Rz[:, 1, 1], Rz[:, 2, 2] = cy, -sy, sy, cy, 1.0
    body = np.einsum("pij,kj->pki", Rz, CANONICAL)  # [P,17,3]
    if cfg.joint_offset is not None:
        body = body + np.asarray(cfg.joint_offset, np.float64).reshape(P, NUM_KEYPOINTS, 3)
    frames, gt_joints, gt_ids = [], [], []
    for f in range(T):
        base = np.concatenate([pos + vel * (f * cfg.dt), np.zeros((P, 1))], 1)
        X = body + base[:, None, :] + rng.normal(0, cfg.joint_jitter, (P, NUM_KEYPOINTS, 3))
        gt_joints.append(X)
        uv, vis = _visible(rig, X)  # [N,P,17,2], [N,P,17]
        uv = uv + rng.normal(0, cfg.pixel_noise, uv.shape)
        score = rng.uniform(0.5, 0.95, vis.shape)
        low = rng.random(vis.shape) < cfg.low_score_frac
        score = np.where(low, rng.uniform(0.0, 0.29, vis.shape), score)
        if cfg.outlier_frac > 0:
            out = rng.random(vis.shape) < cfg.outlier_frac
            sign = np.where(rng.random(uv.shape) < 0.5, -1.0, 1.0)
            uv = np.where(out[..., None], uv + sign * cfg.outlier_px, uv)
        rho = rng.uniform(-0.3, 0.3, vis.shape)
        s2 = cfg.pixel_noise ** 2 if cfg.pixel_noise > 0 else 4.0
        missed = rng.random((N, P)) < cfg.p_miss
        emit = (vis.sum(-1) >= cfg.min_visible_joints) & ~missed  # [N,P]
        det_offset = np.zeros(N + 1, np.int32)
        persons, ids = [], []
        for c in range(N):
            idx = np.nonzero(emit[c])[0]
            if cfg.shuffle:
                idx = rng.permutation(idx)
            arr = np.zeros(len(idx), person2d_dt)
            if len(idx):  # all detections of the camera at once (256 cameras x 128 persons: 32 k detections per frame)
                v = vis[c, idx]                      # [n, 17]
                kp = arr["keypoints"]
                kp["x"] = np.where(v, uv[c, idx, :, 0], 0.0)
                kp["y"] = np.where(v, uv[c, idx, :, 1], 0.0)
                kp["score"] = np.where(v, score[c, idx], 0.0)
                kp["cov"][..., 0] = np.where(v, s2, 0.0)
                kp["cov"][..., 1] = np.where(v, rho[c, idx] * s2, 0.0)
                kp["cov"][..., 2] = np.where(v, s2, 0.0)
                big = np.float32(3e38)
                xs, ys = kp["x"], kp["y"]
                bbox = np.stack([np.where(v, xs, big).min(1), np.where(v, ys, big).min(1), np.where(v, xs, -big).max(1), np.where(v, ys, -big).max(1)], 1)
                anyv = v.any(1)
                arr["bbox"] = np.where(anyv[:, None], bbox, 0.0)
                for k in np.nonzero(anyv)[0]:        # float32 mean over the visible joints only (summation order of the compacted array)
                    arr[k]["score"] = kp["score"][k][v[k]].mean()
            persons.append(arr)
            ids.append(idx)
            det_offset[c + 1] = det_offset[c] + len(idx)
        frames.append(Frame(N, det_offset, np.concatenate(persons) if persons else np.zeros(0, person2d_dt), None, cfg.t0 + f * cfg.dt))
        gt_ids.append(ids)
    return frames, {"joints": np.stack(gt_joints), "ids": gt_ids, "positions": pos, "velocity": vel}
