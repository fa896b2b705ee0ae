"""GPU parity tests for stage 2 (pose_prior): HIP track association + LM (fp64 MFMA J^T J) + marginals + prediction +
track lifecycle vs the CPU oracle on identical measurements.  Refined joints < 1 mm, ids/track counts exact."""
import os
import sys

import numpy as np
import pytest

from smartedgesensor3dhumanpose_amd import person_msgs as pm
from smartedgesensor3dhumanpose_amd import rig as R
from smartedgesensor3dhumanpose_amd import scene as S

sys.path.insert(0, os.path.dirname(__file__))
from parity import compare_stage2  # noqa: E402

pytestmark = pytest.mark.gpu


def _node(rig, **kw):
    from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
    kw.setdefault("max_epi_dist", 0.045)
    return Skeleton3D(rig, **kw)


def _sequence_parity(node, rig, oracle, params, cfg, seq=0, drop_frames=()):
    from smartedgesensor3dhumanpose_amd.pose_prior import PosePrior
    frames, _ = S.generate_sequence(rig, cfg)
    res1 = node.triangulate_persons(frames[: node.limits.max_frames]) if len(frames) <= node.limits.max_frames else None
    node.refine_reset()
    pp = PosePrior(node, seq)
    ora = oracle.PosePrior(params)
    worst = {"pos_max_mm": 0.0, "pred_max_mm": 0.0, "cov_rel_max": 0.0}
    published = 0
    for k, fr in enumerate(frames):
        r1 = res1[k] if res1 is not None else node.triangulate_persons([fr])[0]
        dets = r1.published() if k not in drop_frames else np.zeros(0, pm.person3d_dt)
        g = pp.skeleton_callback(dets, fr.stamp, fr.fb_delay)
        o = ora.step(dets, rig.n_cam, fr.stamp, fr.fb_delay, capacity=node.limits.max_hyp)
        st = compare_stage2(g, o)
        published += g.n_persons
        for key in worst:
            worst[key] = max(worst[key], st[key])
    return worst, published


def test_stage2_parity_16cam_6p_sequence(extr_rig, oracle, demo_params):
    node = _node(extr_rig, max_frames=16, max_det_per_cam=8, max_hyp=32)
    worst, published = _sequence_parity(node, extr_rig, oracle, demo_params, S.SceneConfig(n_persons=6, n_frames=16, seed=0))
    assert published == 6 * 6  # frames 11..16 publish (num_obs > 10)
    print("stage-2 worst:", worst)
    node.close()


def test_stage2_ragged_tracks_appear_vanish_timeout(extr_rig, oracle, demo_params):
    node = _node(extr_rig, max_frames=1, max_det_per_cam=8, max_hyp=32)
    cfg = S.SceneConfig(n_persons=5, n_frames=24, seed=3, p_miss=0.35, low_score_frac=0.15, dt=0.1)
    worst, published = _sequence_parity(node, extr_rig, oracle, demo_params, cfg, drop_frames=(5, 6, 14, 15, 16, 17, 18, 19, 20, 21, 22, 23))
    node.close()


def test_stage2_norm_height_and_h36m(extr_rig, oracle):
    for pose, nh in (("simple", True), ("h36m", False), ("h36m", True)):
        params = pm.Params(pm.POSE_H36M if pose == "h36m" else pm.POSE_SIMPLE, int(nh), 0.045)
        node = _node(extr_rig, pose_method=pose, norm_height=nh, max_frames=12, max_det_per_cam=8, max_hyp=32)
        _sequence_parity(node, extr_rig, oracle, params, S.SceneConfig(n_persons=3, n_frames=12, seed=6))
        node.close()


def test_stage2_resident_pipeline_many_sequences(extr_rig, oracle, demo_params):
    """Resident path: S independent sequences advance one frame per call; stage 2 consumes stage-1 output in HBM."""
    S_ = 5
    T_ = 13
    node = _node(extr_rig, max_frames=S_, max_det_per_cam=8, max_hyp=32)
    seqs = [S.generate_sequence(extr_rig, S.SceneConfig(n_persons=3 + (s % 3), n_frames=T_, seed=40 + s))[0] for s in range(S_)]
    oras = [oracle.PosePrior(demo_params) for _ in range(S_)]
    node.refine_reset()
    total = 0
    for k in range(T_):
        frames = [seqs[s][k] for s in range(S_)]
        node.stage(frames)
        node.run_stage1()
        node.run_stage2([f.stamp for f in frames])
        r1 = node.fetch_stage1(S_)
        r2 = node.fetch_stage2(S_)
        for s in range(S_):
            o = oras[s].step(r1[s].published(), 16, frames[s].stamp, frames[s].fb_delay, capacity=32)
            compare_stage2(r2[s], o)
            total += r2[s].n_persons
    assert total == sum(3 + (s % 3) for s in range(S_)) * (T_ - 10)
    node.close()


def test_stage2_feedback_delay_moving_average(extr_rig, oracle, demo_params):
    node = _node(extr_rig, max_frames=1, max_det_per_cam=8, max_hyp=32)
    frames, _ = S.generate_sequence(extr_rig, S.SceneConfig(n_persons=2, n_frames=5, seed=2))
    rng = np.random.default_rng(0)
    for fr in frames:
        fr.fb_delay[:] = np.where(rng.random(16) < 0.5, rng.uniform(0.05, 0.2, 16), -1.0).astype(np.float32)
    from smartedgesensor3dhumanpose_amd.pose_prior import PosePrior
    node.refine_reset()
    pp, ora = PosePrior(node, 0), oracle.PosePrior(demo_params)
    for fr in frames:
        r1 = node.triangulate_persons([fr])[0]
        g = pp.skeleton_callback(r1.published(), fr.stamp, fr.fb_delay)
        o = ora.step(r1.published(), 16, fr.stamp, fr.fb_delay, capacity=32)
        compare_stage2(g, o)
        assert abs(g.c.pred_delta_t - o.c.pred_delta_t) < 1e-7
    node.close()


def test_device_pack_matches_host_fetch(extr_rig, device_buffer):
    """se3d_pack_stage{1,2}_device (the source of the RCCL all-gather) delivers exactly what the host fetch returns."""
    from smartedgesensor3dhumanpose_amd import shard
    S_, H = 3, 32
    node = _node(extr_rig, max_frames=S_, max_det_per_cam=8, max_hyp=H)
    seqs = [S.generate_sequence(extr_rig, S.SceneConfig(n_persons=3 + s, n_frames=12, seed=90 + s))[0] for s in range(S_)]
    node.refine_reset()
    psz = pm.person3d_dt.itemsize
    p1, c1, p2, c2 = device_buffer(S_ * H * psz), device_buffer(4 * S_), device_buffer(S_ * H * psz), device_buffer(4 * S_)
    for k in range(12):
        frames = [seqs[s][k] for s in range(S_)]
        node.stage(frames)
        node.run_stage1()
        node.run_stage2([f.stamp for f in frames])
        node.pack_stage1_device(p1.data_ptr(), c1.data_ptr())
        node.pack_stage2_device(S_, p2.data_ptr(), c2.data_ptr())
        node.sync()
    r1, r2 = node.fetch_stage1(S_), node.fetch_stage2(S_)
    u1 = shard.unpack_skeletons(p1.numpy(np.uint8).reshape(1, -1), c1.numpy(np.int32).reshape(1, -1), H)[0]
    u2 = shard.unpack_skeletons(p2.numpy(np.uint8).reshape(1, -1), c2.numpy(np.int32).reshape(1, -1), H)[0]
    for s in range(S_):
        assert u1[s].tobytes() == r1[s].published().tobytes()
        assert u2[s].tobytes() == r2[s].fused[: r2[s].n_persons].tobytes()
        assert r2[s].n_persons == 3 + s  # frame 12: num_obs = 12 > 10
    node.close()


def _close_persons(x, y, n):
    """same ids / scores / measured joints; joints within 1e-9 m, covariances within 1e-6 relative (different summation order)"""
    for f in ("id", "score"):
        assert np.array_equal(x[f][:n], y[f][:n])
    kx, ky = x["keypoints"][:n], y["keypoints"][:n]
    assert np.array_equal(kx["score"], ky["score"])
    assert np.allclose(kx["joint"], ky["joint"], rtol=0, atol=1e-9)
    assert np.allclose(kx["cov"], ky["cov"], rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize("S_,T_,persons,mh,rough", [(5, 14, 4, 16, False), (3, 16, 6, 8, False), (2, 12, 9, 64, False), (4, 16, 5, 12, True)])
def test_stage2_walk_equals_per_step_and_oracle(extr_rig, oracle, demo_params, S_, T_, persons, mh, rough):
    """se3d_run_stage2_walk (one fused launch for all time steps; falls back to per-step kernels for max_hyp = 64) against T
    consecutive se3d_run_stage2 calls and the oracle; walked forward then backward (bench.py's ping-pong order).  Variants of the
    fused kernel: one person per LM wavefront ("solo": the per-step device code, bitwise equal to the per-step calls), three persons
    per LM wavefront in the workgroup shape and in the one-wavefront-per-sequence shape (same arithmetic in another summation
    order: bitwise equal to each other, equal decisions and iteration counts, joints within 1e-9 m of the per-step calls)."""
    # rough: cameras that miss persons, many low-score joints and gross 2-D outliers, so that persons reach stage 2 with joints missing
    # (bones inactive, joints that become roots of their own trees, tracks that appear late)
    extra = dict(p_miss=0.08, low_score_frac=0.22, outlier_frac=0.02, pixel_noise=3.0) if rough else {}
    seqs = [S.generate_sequence(extr_rig, S.SceneConfig(n_persons=persons - (s % 2), n_frames=T_, seed=300 + s, **extra))[0] for s in range(S_)]
    frames = [seqs[s][k] for k in range(T_) for s in range(S_)]   # frame slot = k * S + s
    variants = {"solo": dict(walk_lm="solo"), "trio_wg": dict(walk_shape="wg"), "trio_wave": dict(walk_shape="wave")}
    nodes = {name: _node(extr_rig, max_frames=S_ * T_, max_det_per_cam=min(10, mh), max_hyp=mh, options=variants.get(name)) for name in list(variants) + ["steps"]}
    for nd in nodes.values():
        nd.stage(frames)
        nd.run_stage1()
        nd.refine_reset()
    b = nodes["steps"]
    r1 = b.fetch_stage1(S_ * T_)
    order = list(range(T_)) + list(range(T_ - 1, -1, -1))
    t_of = [100.0 + (i + 1) / 30.0 for i in range(len(order))]
    oras = [oracle.PosePrior(demo_params) for _ in range(S_)]
    total = 0
    for half in range(2):
        ks = order[half * T_:(half + 1) * T_]
        ts = t_of[half * T_:(half + 1) * T_]
        for name in variants:
            nodes[name].run_stage2_walk(np.repeat(np.array(ts)[:, None], S_, 1), [k * S_ for k in ks])
            nodes[name].sync()
        for k, t in zip(ks, ts):
            b.run_stage2(np.full(S_, t), frame_slot0=k * S_)
            rb = b.fetch_stage2(S_)
            ra = {name: nodes[name].fetch_stage2_walk(k * S_, S_) for name in variants}
            for s in range(S_):
                n = rb[s].n_persons
                for name in variants:
                    assert ra[name][s].n_persons == n
                    assert np.array_equal(ra[name][s].lm_iterations[:n], rb[s].lm_iterations[:n])
                    assert abs(ra[name][s].c.pred_delta_t - rb[s].c.pred_delta_t) == 0
                assert ra["solo"][s].fused[:n].tobytes() == rb[s].fused[:n].tobytes()
                assert ra["solo"][s].fused_pred[:n].tobytes() == rb[s].fused_pred[:n].tobytes()
                assert ra["trio_wg"][s].fused[:n].tobytes() == ra["trio_wave"][s].fused[:n].tobytes()
                assert ra["trio_wg"][s].fused_pred[:n].tobytes() == ra["trio_wave"][s].fused_pred[:n].tobytes()
                _close_persons(ra["trio_wave"][s].fused, rb[s].fused, n)
                _close_persons(ra["trio_wave"][s].fused_pred, rb[s].fused_pred, n)
                o = oras[s].step(r1[k * S_ + s].published(), 16, t, seqs[s][k].fb_delay, capacity=mh)
                last = k == ks[-1]   # the walk's track count is the state after its last step
                compare_stage2(ra["trio_wave"][s], o, check_tracks=last)
                compare_stage2(ra["trio_wg"][s], o, check_tracks=last)
                assert not last or rb[s].c.n_tracks == o.c.n_tracks
                total += n
    assert total > S_ * T_ * (persons - 1) or rough
    if rough:   # the rough scene must actually exercise incomplete skeletons
        r1all = b.fetch_stage1(S_ * T_)
        nj = np.concatenate([(r.published()["keypoints"]["score"] > 0.10).sum(1) for r in r1all if r.n_persons])
        assert nj.min() < nj.max() and total > S_ * T_
    for nd in nodes.values():
        nd.close()


@pytest.mark.parametrize("pose,nh", [("simple", True), ("h36m", False), ("h36m", True)])
def test_stage2_walk_norm_height_and_h36m(extr_rig, oracle, pose, nh):
    """The fused walk (three persons per LM wavefront, both kernel shapes) with height normalisation and with the H36M joint set,
    where the belly is measured and the bone forest is rooted there: against the per-step kernels and the oracle."""
    params = pm.Params(pm.POSE_H36M if pose == "h36m" else pm.POSE_SIMPLE, int(nh), 0.045)
    S_, T_, mh = 3, 14, 8
    seqs = [S.generate_sequence(extr_rig, S.SceneConfig(n_persons=4 - (s % 2), n_frames=T_, seed=900 + s))[0] for s in range(S_)]
    frames = [seqs[s][k] for k in range(T_) for s in range(S_)]
    nodes = {name: _node(extr_rig, pose_method=pose, norm_height=nh, max_frames=S_ * T_, max_det_per_cam=8, max_hyp=mh) for name in ("wg", "wave", "steps")}
    for nd in nodes.values():
        nd.stage(frames)
        nd.run_stage1()
        nd.refine_reset()
    b = nodes["steps"]
    r1 = b.fetch_stage1(S_ * T_)
    ts = [100.0 + (i + 1) / 30.0 for i in range(T_)]
    for name in ("wg", "wave"):
        nodes[name].set_options(walk_shape=name)
        nodes[name].run_stage2_walk(np.repeat(np.array(ts)[:, None], S_, 1), [k * S_ for k in range(T_)])
        nodes[name].sync()
    oras = [oracle.PosePrior(params) for _ in range(S_)]
    total = 0
    for k, t in enumerate(ts):
        b.run_stage2(np.full(S_, t), frame_slot0=k * S_)
        rb = b.fetch_stage2(S_)
        ra = {name: nodes[name].fetch_stage2_walk(k * S_, S_) for name in ("wg", "wave")}
        for s in range(S_):
            n = rb[s].n_persons
            o = oras[s].step(r1[k * S_ + s].published(), 16, t, seqs[s][k].fb_delay, capacity=mh)
            for name in ("wg", "wave"):
                assert ra[name][s].n_persons == n
                assert np.array_equal(ra[name][s].lm_iterations[:n], rb[s].lm_iterations[:n])
                _close_persons(ra[name][s].fused, rb[s].fused, n)
                _close_persons(ra[name][s].fused_pred, rb[s].fused_pred, n)
                compare_stage2(ra[name][s], o, check_tracks=k == T_ - 1)
            assert ra["wg"][s].fused[:n].tobytes() == ra["wave"][s].fused[:n].tobytes()
            total += n
    assert total > 0
    for nd in nodes.values():
        nd.close()


@pytest.mark.parametrize("walk", [False, True])
def test_stage2_track_capacity_flag(extr_rig, walk):
    """Five persons on a context with room for three tracks: the overflow is reported (sticky, per sequence) instead of passing silently;
    a second sequence with two persons stays clean.  Per-step kernels and the fused walk."""
    T_ = 4
    seqs = [S.generate_sequence(extr_rig, S.SceneConfig(n_persons=n, n_frames=T_, seed=20 + n))[0] for n in (5, 2)]
    frames = [seqs[s][k] for k in range(T_) for s in range(2)]
    node = _node(extr_rig, max_frames=2 * T_, max_det_per_cam=8, max_hyp=8, max_tracks=3)
    node.stage(frames)
    node.run_stage1()
    node.refine_reset()
    ts = [100.0 + (i + 1) / 30.0 for i in range(T_)]
    if walk:
        node.run_stage2_walk(np.repeat(np.array(ts)[:, None], 2, 1), [k * 2 for k in range(T_)])
        r = node.fetch_stage2_walk((T_ - 1) * 2, 2)
    else:
        for k in range(T_):
            node.run_stage2(np.full(2, ts[k]), frame_slot0=k * 2)
        r = node.fetch_stage2(2)
    assert r[0].c.flags & pm.S2FLAG_TRACK_OVERFLOW and r[0].c.n_tracks == 3
    assert r[1].c.flags == 0 and r[1].c.n_tracks == 2
    node.refine_reset()
    node.run_stage2(np.full(2, 200.0), frame_slot0=0)
    assert node.fetch_stage2(2)[1].c.flags == 0
    node.close()


@pytest.mark.parametrize("shape", ["wave", "wg"])
def test_stage2_walk_is_deterministic_and_sequences_do_not_interact(extr_rig, shape):
    """Size-independent properties of the fused walk: the same batch twice gives the same bytes, and a sequence's results do not depend
    on which other sequences share the launch (reversed sequence order, i.e. other workgroup / wavefront neighbours and LDS regions)."""
    S_, T_ = 7, 14
    seqs = [S.generate_sequence(extr_rig, S.SceneConfig(n_persons=2 + (s % 5), n_frames=T_, seed=700 + s, p_miss=0.05 * (s % 3)))[0] for s in range(S_)]
    ts = [100.0 + (i + 1) / 30.0 for i in range(T_)]
    outs = []
    for order in (list(range(S_)), list(range(S_)), list(range(S_ - 1, -1, -1))):
        frames = [seqs[s][k] for k in range(T_) for s in order]
        node = _node(extr_rig, max_frames=S_ * T_, max_det_per_cam=8, max_hyp=12, options=dict(walk_shape=shape))
        node.stage(frames)
        node.run_stage1()
        node.refine_reset()
        node.run_stage2_walk(np.repeat(np.array(ts)[:, None], S_, 1), [k * S_ for k in range(T_)])
        res = {}
        for k in range(T_):
            r = node.fetch_stage2_walk(k * S_, S_)
            for pos, s in enumerate(order):
                n = r[pos].n_persons
                res[(s, k)] = (n, r[pos].fused[:n].tobytes(), r[pos].fused_pred[:n].tobytes(), r[pos].lm_iterations[:n].tobytes())
        outs.append(res)
        node.close()
    assert outs[0] == outs[1], "run-to-run difference"
    assert outs[0] == outs[2], "a sequence's results depend on its neighbours in the launch"
    assert sum(v[0] for v in outs[0].values()) > S_ * 3


def test_stage2_walk_bookkeeping_in_lds_equals_global(extr_rig):
    """One-wavefront walk: the per-step bookkeeping (cost matrix, merge rows, order, slot / publish tables, per-sequence scalars) in LDS
    (default when it fits) against the same tables in global memory (options.walk_book = 1): published persons, iteration counts, track counts and
    flags byte for byte, over a ping-pong walk of sequences with appearing / vanishing persons; then the per-step kernels continue from the
    state both left behind (order, n_tracks, next_id, frame_nr, t_prev, feedback buffer were written back)."""
    S_, T_ = 5, 12
    seqs = [S.generate_sequence(extr_rig, S.SceneConfig(n_persons=3 + (s % 4), n_frames=T_, seed=900 + s, p_miss=0.04 * (s % 3), low_score_frac=0.1))[0] for s in range(S_)]
    frames = [seqs[s][k] for k in range(T_) for s in range(S_)]
    outs = []
    for book in (0, 1):
        node = _node(extr_rig, max_frames=S_ * T_, max_det_per_cam=8, max_hyp=12, options=dict(walk_shape="wave", walk_book=book))
        node.stage(frames)
        node.run_stage1()
        node.refine_reset()
        t = 100.0
        res = []
        for ks in (list(range(T_)), list(range(T_ - 1, -1, -1))):
            ts = t + np.arange(1, T_ + 1) / 30.0
            t = float(ts[-1])
            node.run_stage2_walk(np.repeat(ts[:, None], S_, 1), [k * S_ for k in ks])
            for k in ks:
                for r in node.fetch_stage2_walk(k * S_, S_):
                    n = r.n_persons
                    res.append((n, r.fused[:n].tobytes(), r.fused_pred[:n].tobytes(), r.lm_iterations[:n].tobytes(), r.c.n_tracks, r.c.flags))
        node.run_stage2(np.full(S_, t + 1 / 30.0), frame_slot0=0)   # continue with the per-step kernels from the written-back state
        for r in node.fetch_stage2(S_):
            n = r.n_persons
            res.append((n, r.fused[:n].tobytes(), r.lm_iterations[:n].tobytes(), r.c.n_tracks))
        outs.append(res)
        node.close()
    assert outs[0] == outs[1]
    assert sum(v[0] for v in outs[0]) > S_ * T_


def test_stage2_walk_at_bench_size(extr_rig, oracle, demo_params):
    """The bench.py shape of 16 cam x 6 persons: 256 sequences x 16 frames in one launch (the launcher picks the one-wavefront-per-sequence
    kernel from 256 sequences).  The distinct sequences are checked against the oracle step by step, every replica must be bitwise
    equal to its prototype, and the steady-state pass publishes every person of every frame."""
    import bench
    w = bench.WORKLOADS["16cam6p"]
    S_, T_, U = 256, 16, 32
    seqs = bench.make_sequences(None, w, 1, S_, 4, T_)[0]   # one context: up to 32 distinct seeded sequences (4000 ...), repeated
    frames = [seqs[s][k] for k in range(T_) for s in range(S_)]
    node = _node(extr_rig, max_frames=S_ * T_, max_det_per_cam=w["max_det"], max_hyp=w["max_hyp"])
    node.stage(frames)
    node.run_stage1()
    node.refine_reset()
    r1 = node.fetch_stage1(S_ * T_)
    oras = [oracle.PosePrior(demo_params) for _ in range(U)]
    t = 100.0
    for rep in range(2):   # forward, then backward (bench.py's ping-pong): the second pass is in steady state
        ks = list(range(T_)) if rep == 0 else list(range(T_ - 1, -1, -1))
        ts = t + np.arange(1, T_ + 1) / 30.0
        t = float(ts[-1])
        node.run_stage2_walk(np.repeat(ts[:, None], S_, 1), [k * S_ for k in ks])
        published = 0
        for k, tk in zip(ks, ts):
            r = node.fetch_stage2_walk(k * S_, S_)
            for u in range(U):
                o = oras[u].step(r1[k * S_ + u].published(), 16, float(tk), seqs[u][k].fb_delay, capacity=w["max_hyp"])
                compare_stage2(r[u], o, check_tracks=(k == ks[-1]))
            for s in range(S_):
                n = r[s].n_persons
                assert n == r[s % U].n_persons and r[s].fused[:n].tobytes() == r[s % U].fused[:n].tobytes()
                published += n
    assert published == S_ * T_ * w["n_persons"]
    node.close()


def test_stage2_walk_device_pack(extr_rig, device_buffer):
    from smartedgesensor3dhumanpose_amd import shard
    S_, T_, H = 4, 12, 8
    seqs = [S.generate_sequence(extr_rig, S.SceneConfig(n_persons=5, n_frames=T_, seed=70 + s))[0] for s in range(S_)]
    frames = [seqs[s][k] for k in range(T_) for s in range(S_)]
    node = _node(extr_rig, max_frames=S_ * T_, max_det_per_cam=8, max_hyp=H)
    node.stage(frames)
    node.run_stage1()
    node.refine_reset()
    psz = pm.person3d_dt.itemsize
    p, c = device_buffer(S_ * T_ * H * psz), device_buffer(4 * S_ * T_)
    for rep in range(2):
        ks = list(range(T_)) if rep == 0 else list(range(T_ - 1, -1, -1))
        ts = [100.0 + (rep * T_ + i + 1) / 30.0 for i in range(T_)]
        node.run_stage2_walk(np.repeat(np.array(ts)[:, None], S_, 1), [k * S_ for k in ks])
    node.pack_stage2_walk_device(S_ * T_, p.data_ptr(), c.data_ptr())
    XH = 6   # compact exchange layout: 6 of the 8 person slots per frame
    p6, c6 = device_buffer(S_ * T_ * XH * psz), device_buffer(4 * S_ * T_)
    node.pack_stage2_walk_device_slots(S_ * T_, XH, p6.data_ptr(), c6.data_ptr())
    node.sync()
    u = shard.unpack_skeletons(p.numpy(np.uint8).reshape(1, -1), c.numpy(np.int32).reshape(1, -1), H)[0]
    u6 = shard.unpack_skeletons(p6.numpy(np.uint8).reshape(1, -1), c6.numpy(np.int32).reshape(1, -1), XH)[0]
    assert all(a.tobytes() == b.tobytes() for a, b in zip(u, u6))
    # dense exchange layout: the persons back to back + offsets; with a capacity that is too small the tail is dropped and counted
    tot = sum(len(x) for x in u)
    pd, od = device_buffer(tot * psz), device_buffer(4 * (S_ * T_ + 2))
    node.pack_stage2_walk_device_dense(S_ * T_, tot, pd.data_ptr(), od.data_ptr())
    ps, os_ = device_buffer((tot - 7) * psz), device_buffer(4 * (S_ * T_ + 2))
    node.pack_stage2_walk_device_dense(S_ * T_, tot - 7, ps.data_ptr(), os_.data_ptr())
    node.sync()
    ud, dropped = shard.unpack_dense(pd.numpy(np.uint8), od.numpy(np.int32), S_ * T_)
    assert dropped == 0 and all(a.tobytes() == b.tobytes() for a, b in zip(u, ud))
    hp, ho = shard.pack_host_dense(u, tot)   # the host twin the gloo tests use writes the same bytes
    assert hp.numpy().tobytes() == pd.numpy(np.uint8).tobytes() and ho.numpy().tolist() == od.numpy(np.int32).tolist()
    us, dropped = shard.unpack_dense(ps.numpy(np.uint8), os_.numpy(np.int32), S_ * T_)
    assert dropped == 7 and sum(len(x) for x in us) == tot - 7
    assert b"".join(x.tobytes() for x in us) == b"".join(x.tobytes() for x in u)[: (tot - 7) * psz]
    hp, ho = shard.pack_host_dense(u, tot - 7)
    assert ho.numpy().tolist() == os_.numpy(np.int32).tolist()
    got = 0
    for k in range(T_):
        r = node.fetch_stage2_walk(k * S_, S_)
        for s in range(S_):
            n = r[s].n_persons
            assert len(u[k * S_ + s]) == n
            assert u[k * S_ + s].tobytes() == r[s].fused[:n].tobytes()
            got += n
    assert got == S_ * T_ * 5
    node.close()


@pytest.mark.parametrize("variant", ["solo", "trio"])
def test_stage2_walk_then_per_step_share_track_state(extr_rig, variant):
    """Track state written back by a fused walk (it lives in LDS during the walk) is what later per-step calls continue from:
    bitwise with the one-person-per-wavefront LM (the per-step device code), within rounding with three persons per wavefront."""
    S_, T_ = 3, 16
    seqs = [S.generate_sequence(extr_rig, S.SceneConfig(n_persons=5, n_frames=T_, seed=500 + s))[0] for s in range(S_)]
    frames = [seqs[s][k] for k in range(T_) for s in range(S_)]
    a = _node(extr_rig, max_frames=S_ * T_, max_det_per_cam=8, max_hyp=8, options=dict(walk_lm=variant))
    b = _node(extr_rig, max_frames=S_ * T_, max_det_per_cam=8, max_hyp=8, options=dict(walk_lm=variant))
    for nd in (a, b):
        nd.stage(frames)
        nd.run_stage1()
        nd.refine_reset()
    ts = [100.0 + (i + 1) / 30.0 for i in range(T_)]
    half = T_ // 2
    a.run_stage2_walk(np.repeat(np.array(ts[:half])[:, None], S_, 1), [k * S_ for k in range(half)])
    for k in range(half):
        b.run_stage2(np.full(S_, ts[k]), frame_slot0=k * S_)
    published = 0
    for k in range(half, T_):
        for nd in (a, b):
            nd.run_stage2(np.full(S_, ts[k]), frame_slot0=k * S_)
        ra, rb = a.fetch_stage2(S_), b.fetch_stage2(S_)
        for s in range(S_):
            n = ra[s].n_persons
            assert n == rb[s].n_persons and ra[s].c.n_tracks == rb[s].c.n_tracks
            if variant == "solo":
                assert ra[s].fused[:n].tobytes() == rb[s].fused[:n].tobytes()
                assert ra[s].fused_pred[:n].tobytes() == rb[s].fused_pred[:n].tobytes()
            else:
                _close_persons(ra[s].fused, rb[s].fused, n)
                _close_persons(ra[s].fused_pred, rb[s].fused_pred, n)
            published += n
    assert published > 0
    a.close()
    b.close()


LM_SCENES = {
    # the per-step LM kernel under each normal-equation form against the oracle: joints < 1 mm, iteration counts equal,
    # marginal covariances rtol 1e-4 — on a full 16-frame sequence, a ragged one (tracks appear, vanish, time out) and the h36m /
    # height-normalised graph (21 variables, the other bone table)
    "six_persons": dict(cfg=dict(n_persons=6, n_frames=16, seed=0), pose="simple", nh=False, drop=()),
    "ragged": dict(cfg=dict(n_persons=5, n_frames=20, seed=3, p_miss=0.35, low_score_frac=0.15, dt=0.1), pose="simple", nh=False, drop=(5, 6, 14, 15, 16)),
    "h36m_norm_height": dict(cfg=dict(n_persons=4, n_frames=14, seed=6), pose="h36m", nh=True, drop=()),
    "outliers": dict(cfg=dict(n_persons=3, n_frames=14, seed=11, outlier_frac=0.04), pose="simple", nh=False, drop=()),
}


@pytest.mark.parametrize("scene", sorted(LM_SCENES))
@pytest.mark.parametrize("lm", ["tree", "tree_mfma", "dense"])
def test_stage2_every_lm_form_matches_the_oracle(extr_rig, oracle, lm, scene):
    """north_star config 4: the batched J^T J on the matrix cores (`dense`: v_mfma_f64_16x16x4_f64 on the whole 63 x 63 product;
    `tree_mfma`: on the bone-factor block of the marginals) and the default forest elimination, each through se3d_refine against the oracle on
    four scenes (the per-step LM kernel: refine_steps = 1)."""
    sc = LM_SCENES[scene]
    params = pm.Params(pm.POSE_H36M if sc["pose"] == "h36m" else pm.POSE_SIMPLE, int(sc["nh"]), 0.045)
    node = _node(extr_rig, pose_method=sc["pose"], norm_height=sc["nh"], max_frames=1, max_det_per_cam=8, max_hyp=32, options=dict(lm_form=lm, refine_steps=1))
    worst, published = _sequence_parity(node, extr_rig, oracle, params, S.SceneConfig(**sc["cfg"]), drop_frames=sc["drop"])
    print("lm_form=%s on %s: %d published persons, worst %s" % (lm, scene, published, worst))
    assert published > 0
    node.close()


@pytest.mark.parametrize("n_cam,n_persons", [(32, 12), (48, 16)])
def test_stage2_large_rig_every_step_asserted(oracle, n_cam, n_persons):
    """A clean scene on a 32 / 48-camera rig (the per-step stage-2 kernels with the block Munkres, H = 64): EVERY sequence-step is compared
    with the oracle — ids, track counts, LM iteration counts, refined and predicted joints, marginal covariances; nothing is exempt."""
    from smartedgesensor3dhumanpose_amd.skeleton_3d import rig_tensors
    rig = R.elliptical_rig(n_cam, seed=2)
    rig.set_tensors(*rig_tensors(rig))
    params = pm.Params(pm.POSE_SIMPLE, 0, 0.045)
    node = _node(rig, max_frames=1, max_det_per_cam=24, max_hyp=64)
    worst, published = _sequence_parity(node, rig, oracle, params, S.SceneConfig(n_persons=n_persons, n_frames=14, seed=21, spacing=1.0))
    print("%d cameras x %d persons: %d published persons over 14 steps, worst %s" % (n_cam, n_persons, published, worst))
    assert published >= 3 * n_persons - 2
    node.close()


def test_pose_prior_state_moves_between_contexts_and_matches_the_oracles_tracks(extr_rig, oracle, demo_params):
    """PosePrior.save_state / load_state / tracks (se3d_refine_save / _load): after eight frames the node's track table — ids, observation
    counts, previous root / height, previous estimates — equals the oracle's g_tracks; the blob loaded into a context with LARGER limits makes
    that context hand the same bytes back and continue the sequence with the first context's outputs (to 1e-9 m: the two limits select different
    LM kernel forms), and both stay on the oracle."""
    from smartedgesensor3dhumanpose_amd.pose_prior import PosePrior
    from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
    frames, _ = S.generate_sequence(extr_rig, S.SceneConfig(n_persons=5, n_frames=16, seed=21, p_miss=0.15))
    a = Skeleton3D(extr_rig, max_epi_dist=0.045, max_frames=1, max_det_per_cam=8, max_hyp=12, device=0)
    b = Skeleton3D(extr_rig, max_epi_dist=0.045, max_frames=1, max_det_per_cam=8, max_hyp=32, max_tracks=48, device=0)
    a.refine_reset(); b.refine_reset()
    pa, pb, ora = PosePrior(a, 0), PosePrior(b, 0), oracle.PosePrior(demo_params)
    dets = [a.triangulate_persons([fr])[0].published() for fr in frames]
    for k in range(8):
        compare_stage2(pa.skeleton_callback(dets[k], frames[k].stamp, frames[k].fb_delay), ora.step(dets[k], 16, frames[k].stamp, frames[k].fb_delay, capacity=32))
    tr, ot = pa.tracks(), ora.tracks()
    assert tr["id"].tolist() == ot["ids"].tolist() and tr["num_obs"].tolist() == ot["num_obs"].tolist() and len(tr) >= 5
    assert np.array_equal(tr["mask"], ot["mask"]) and np.allclose(tr["t_prev"], ot["t_prev"])
    assert np.abs(tr["root_prev"] - ot["root_prev"]).max() < 1e-9 and np.abs(tr["height_prev"] - ot["height_prev"]).max() < 1e-12
    held = ((tr["mask"][:, None] >> np.arange(pm.NUM_FUSION_KEYPOINTS)[None, :]) & 1).astype(bool)   # joints in prevEstimate (the other slots of the blob are undefined)
    assert held.sum() > 50 and np.abs(tr["prev"] - ot["prev"])[held].max() < 1e-6      # the previous estimates (normalised joints) of every track
    blob = pa.save_state()
    hd, _ = PosePrior.parse_state(blob)
    assert int(hd["n_tracks"]) == len(tr) and int(hd["frame_nr"]) == 8
    pb.load_state(blob)
    assert (tr["prev"][~held] == 0).all() and (tr["vel"][~held] == 0).all()   # joints outside prevEstimate: zeros in the blob
    assert pb.save_state() == blob                             # the same bytes out of the larger context
    published = 0
    for k in range(8, 16):
        ra = pa.skeleton_callback(dets[k], frames[k].stamp, frames[k].fb_delay)
        rb = pb.skeleton_callback(dets[k], frames[k].stamp, frames[k].fb_delay)
        n = ra.n_persons
        assert rb.n_persons == n and ra.c.n_tracks == rb.c.n_tracks
        assert np.array_equal(ra.fused["id"][:n], rb.fused["id"][:n]) and np.array_equal(ra.lm_iterations[:n], rb.lm_iterations[:n])
        for nm in ("fused", "fused_pred") if n else ():   # (limits 12 / 32 select different LM kernel forms: equal up to the summation order)
            ga, gb = getattr(ra, nm)[:n]["keypoints"], getattr(rb, nm)[:n]["keypoints"]
            assert np.array_equal(ga["score"], gb["score"]) and np.abs(ga["joint"] - gb["joint"]).max() < 1e-9
        compare_stage2(rb, ora.step(dets[k], 16, frames[k].stamp, frames[k].fb_delay, capacity=32))
        published += n
    assert published >= 20
    a.close(); b.close()


def _flat_person_run(extr_rig, oracle, demo_params, dist, sigma):
    """16 steps of four clean stage-1 persons, person 0 carrying a joint whose covariance is sigma^2 along a camera ray (branch_scenes.garbage_joint):
    both LMs on the same inputs; returns the tallies of parity.compare_stage2_objectives and the per-person records of the disagreements."""
    import branch_scenes
    from parity import compare_stage2_objectives
    from smartedgesensor3dhumanpose_amd.pose_prior import PosePrior
    from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
    frames, _ = S.generate_sequence(extr_rig, S.SceneConfig(n_persons=4, n_frames=16, seed=3))
    s1 = [oracle.triangulate(extr_rig, demo_params, fr, capacity=32, hyp_capacity=32).published() for fr in frames]
    node = Skeleton3D(extr_rig, max_epi_dist=0.045, max_frames=1, max_det_per_cam=8, max_hyp=16, device=0)
    pp, ora = PosePrior(node, 0), oracle.PosePrior(demo_params)
    tot = dict(persons=0, disagree=0, worse=0, status_differs=0, records=[])
    for k, fr in enumerate(frames):
        dets = branch_scenes.garbage_joint(extr_rig, s1[k], person=0, joint=9, dist=dist, sigma=sigma)
        g = pp.skeleton_callback(dets, fr.stamp, fr.fb_delay)
        o = ora.step(dets, extr_rig.n_cam, fr.stamp, fr.fb_delay, capacity=16)
        oc = compare_stage2_objectives(g, o)
        assert oc["comparable"], "the two sides publish different detections"
        tot["persons"] += oc["persons"]; tot["disagree"] += oc["disagree"]; tot["worse"] += oc["worse"]
        n = g.n_persons
        for p in range(n):
            d = np.linalg.norm(g.fused[p]["keypoints"]["joint"] - o.fused[p]["keypoints"]["joint"], axis=-1)[o.fused[p]["keypoints"]["score"] > 0].max()
            assert np.isfinite(g.fused[p]["keypoints"]["joint"]).all() and np.isfinite(g.fused_pred[p]["keypoints"]["joint"]).all()
            if d >= 1e-3:
                tot["status_differs"] += int((g.lm_objective[p] < 0) != (o.lm_objective[p] < 0))
                tot["records"].append(dict(step=k, person=p, d_mm=round(float(d) * 1e3, 3), objective_lib=float(g.lm_objective[p]), objective_oracle=float(o.lm_objective[p]),
                                           iterations=(int(g.lm_iterations[p]), int(o.lm_iterations[p]))))
    node.close()
    return tot


@pytest.mark.parametrize("dist", [0.0, 0.3])
def test_objective_criterion_fires_on_a_flat_person_and_passes(extr_rig, oracle, demo_params, dist):
    """Round 6 (the solver-independent criterion of round 5 never met a published person whose refined joints differ between the two
    sides, so it proved nothing).  A constructed input on which it does: a PUBLISHED person with one joint known to 300 m along a camera ray and 2 cm
    across — what two nearly parallel rays triangulate to.  The person's LM problem is flat along that ray (unary information 1e-5 m^-2 next to a bone
    factor of ~1e3): the library's forest elimination and the oracle's dense solve take different accept / reject paths (40 against 28 iterations) and
    stop millimetres apart.  parity.compare_stage2_objectives must see the disagreement AND find the library's final objective — the same function
    on both sides, GTSAM's graph.error (PP:746-758) — not the larger one."""
    tot = _flat_person_run(extr_rig, oracle, demo_params, dist, 300.0)
    print("sigma 300 m, dist %.1f m: %s" % (dist, {k: v for k, v in tot.items()}))
    assert tot["persons"] >= 20
    assert tot["disagree"] >= 1, "the constructed scene no longer makes the two LMs stop apart: the criterion is not exercised"
    assert tot["worse"] == 0 and tot["status_differs"] == 0, tot["records"]
    for r in tot["records"]:
        assert 0 < r["objective_lib"] <= r["objective_oracle"] * (1 + 1e-9), r


def test_near_singular_person_is_published_by_both_sides_whatever_the_solver_says(extr_rig, oracle, demo_params):
    """The same joint known to 10 km / 100 km along the ray (information 1e-8 / 1e-10 m^-2, condition of the person's normal equations 1e11 - 1e13) —
    far outside anything the plausibility filters let through, kept as a RECORD of where the two restatements of GTSAM's semantics part: they
    disagree on WHETHER the optimisation failed (PP:751-758: an IndeterminantLinearSystemException sends the raw measurements out with the default
    covariance) — one side's elimination still finds a positive pivot where the other's does not, on a step or two of sixteen.  Which of them GTSAM
    4.0.3's multifrontal Cholesky sides with cannot be settled here (GTSAM is absent: DESIGN section 4).  Asserted: both sides publish the same
    persons with finite joints on every step, and the differing steps are few."""
    for sigma in (1e4, 1e5):
        tot = _flat_person_run(extr_rig, oracle, demo_params, 0.0, sigma)
        print("sigma %.0e m: %s" % (sigma, {k: v for k, v in tot.items()}))
        assert tot["persons"] >= 20 and tot["disagree"] <= 2 and tot["status_differs"] <= 2, tot["records"]
