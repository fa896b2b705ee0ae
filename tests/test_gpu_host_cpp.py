"""GPU test of the C++ host layer (host/se3d_nodes.cpp via host/se3d_replay): the compiled node twins drive libse3d_hip.so
through the C-ABI; their published lists are compared with the oracle pushed through the Python harness twin."""
import os
import subprocess
import sys

import numpy as np
import pytest

from smartedgesensor3dhumanpose_amd import harness as Hn
from smartedgesensor3dhumanpose_amd import scene as S

sys.path.insert(0, os.path.dirname(__file__))
from parity import POS_TOL_M, REFINE_TOL_M  # noqa: E402

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_replay_matches_oracle(tmp_path, oracle, extr_rig, demo_params):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host")], stdout=subprocess.DEVNULL)
    frames, _ = S.generate_sequence(extr_rig, S.SceneConfig(n_persons=5, n_frames=14, seed=12))
    stream = Hn.stream_from_frames(frames)
    stream[3][2].stamp -= 0.100                 # stale view
    stream[6] = [Hn.CameraMsg(m.stamp - 1.0, m.fb_delay, m.persons) for m in stream[6]]   # jumps backwards: dropped
    for m in stream[9]:
        m.fb_delay = 0.08                       # feedback delay measured on the sensors
    scene, out = str(tmp_path / "scene.bin"), str(tmp_path / "out.bin")
    Hn.write_replay_scene(scene, extr_rig, demo_params, stream)
    log = subprocess.run([os.path.join(ROOT, "host", "se3d_replay"), scene, out], check=True, capture_output=True, text=True).stdout
    got = Hn.read_replay_output(out)
    # the MarkerArray publications of the two nodes (host/se3d_markers.hpp): a line list and a sphere list per published person
    marker_line = [ln for ln in log.split("\n") if ln.startswith("Markers:")][0].replace(",", "").split()
    assert int(marker_line[1]) == 2 * sum(len(g[0].persons) for g in got if g is not None)
    assert int(marker_line[3]) == 2 * sum(len(g[1].persons) for g in got if g is not None)

    pp = oracle.PosePrior(demo_params)
    h1 = Hn.Skeleton3DHarness(16, lambda fr: oracle.triangulate(extr_rig, demo_params, fr, capacity=64, hyp_capacity=64))
    h2 = Hn.PosePriorHarness(lambda persons, n_cam, t, fb: pp.step(persons, n_cam, t, fb, capacity=64))
    n_checked = 0
    for k, msgs in enumerate(stream):
        p3d = h1.step(msgs)
        if p3d is None:
            assert got[k] is None
            continue
        g3d, gfused, gpred = got[k]
        assert g3d.ts_per_cam == p3d.ts_per_cam and np.allclose(g3d.fb_delay_per_cam, p3d.fb_delay_per_cam)
        assert len(g3d.persons) == len(p3d.persons)
        v = p3d.persons["keypoints"]["score"] > 0
        assert np.array_equal(g3d.persons["keypoints"]["score"] > 0, v)
        d = np.linalg.norm(g3d.persons["keypoints"]["joint"] - p3d.persons["keypoints"]["joint"], axis=-1)[v]
        assert d.max() < POS_TOL_M
        # stage 2 consumes the HIP stage-1 persons in the C++ pipeline and the oracle's in the reference pipeline: refined joints
        # inherit the (sub-micrometre) stage-1 difference
        ofused, opred = h2.callback(p3d)
        assert len(gfused.persons) == len(ofused.persons)
        if len(ofused.persons):
            assert np.array_equal(gfused.persons["id"], ofused.persons["id"])
            vv = ofused.persons["keypoints"]["score"] > 0
            dd = np.linalg.norm(gfused.persons["keypoints"]["joint"] - ofused.persons["keypoints"]["joint"], axis=-1)[vv]
            assert dd.max() < REFINE_TOL_M
            assert np.allclose(gfused.fb_delay_per_cam, ofused.fb_delay_per_cam, atol=1e-7)
            n_checked += len(ofused.persons)
    assert got[6] is None and n_checked > 0


def test_cpp_replay_reprojection(tmp_path, oracle, extr_rig, demo_params):
    """skeleton_3d -> pose_prior -> pose_reprojection chain of the C++ host layer: the per-camera skel_pred lists equal the
    oracle's reprojection of the predicted skeletons the same run published."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host")], stdout=subprocess.DEVNULL)
    frames, _ = S.generate_sequence(extr_rig, S.SceneConfig(n_persons=4, n_frames=16, seed=5))
    stream = Hn.stream_from_frames(frames)
    scene, out, rp = str(tmp_path / "scene.bin"), str(tmp_path / "out.bin"), str(tmp_path / "reproj.bin")
    Hn.write_replay_scene(scene, extr_rig, demo_params, stream)
    subprocess.check_call([os.path.join(ROOT, "host", "se3d_replay"), scene, out, rp, str(extr_rig.width), str(extr_rig.height)])
    got = [g for g in Hn.read_replay_output(out) if g is not None]
    lists = Hn.read_replay_reproj(rp)
    assert len(lists) == len(got)
    n_checked = 0
    for (p3d, fused, pred), cams in zip(got, lists):
        assert cams is not None and len(cams) == extr_rig.n_cam
        ref = oracle.reproject(extr_rig, "simple", pred.persons)
        for i, (stamp, fb, persons) in enumerate(cams):
            assert stamp == pred.ts_per_cam[i] and abs(fb - pred.fb_delay_per_cam[i]) < 1e-7
            rper, _ = ref.camera(i)
            assert len(persons) == len(rper)
            if len(persons):
                assert np.array_equal(persons["keypoints"]["score"], rper["keypoints"]["score"])
                np.testing.assert_allclose(persons["keypoints"]["x"], rper["keypoints"]["x"], rtol=2e-6, atol=1e-4)
                np.testing.assert_allclose(persons["keypoints"]["cov"], rper["keypoints"]["cov"], rtol=2e-6, atol=1e-4)
                np.testing.assert_allclose(persons["bbox"], rper["bbox"], rtol=2e-6, atol=1e-4)
                n_checked += len(persons)
    assert n_checked > 50


def test_cpp_replay_through_time_synchroniser(tmp_path, extr_rig, demo_params):
    """With SE3D_REPLAY_SYNC=1 the per-camera messages (stamps jittered by a few ms inside each frame) go through
    se3d::Skeleton3DFrontEnd (approximate-time synchroniser, node parameters of S3D:1219-1221) before skeletonThreadStep: every
    tuple it publishes must be the frame itself, so the published 3-D lists equal the directly fed replay."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host")], stdout=subprocess.DEVNULL)
    frames, _ = S.generate_sequence(extr_rig, S.SceneConfig(n_persons=4, n_frames=14, seed=23))
    stream = Hn.stream_from_frames(frames)
    rng = np.random.default_rng(1)
    for msgs in stream:
        for m in msgs:
            m.stamp = round(m.stamp + float(rng.uniform(-0.003, 0.003)), 6)
    scene, out_a, out_b = str(tmp_path / "scene.bin"), str(tmp_path / "a.bin"), str(tmp_path / "b.bin")
    Hn.write_replay_scene(scene, extr_rig, demo_params, stream)
    exe = os.path.join(ROOT, "host", "se3d_replay")
    subprocess.check_call([exe, scene, out_a])
    subprocess.check_call([exe, scene, out_b], env=dict(os.environ, SE3D_REPLAY_SYNC="1"))
    a, b = Hn.read_replay_output(out_a), Hn.read_replay_output(out_b)
    assert len(a) == len(stream) and len(stream) - 2 <= len(b) <= len(stream)   # the synchroniser may still hold the last frames
    for ra, rb in zip(a, b):
        assert (ra is None) == (rb is None)
        if ra is None:
            continue
        for la, lb in zip(ra, rb):
            assert la.stamp == lb.stamp and la.ts_per_cam == lb.ts_per_cam
            assert la.persons.tobytes() == lb.persons.tobytes()


def test_cpp_replay_grows_its_limits_instead_of_truncating(tmp_path, oracle, extr_rig, demo_params):
    """The reference never truncates (S3D:636-673 appends hypotheses, PP:570-580 appends tracks); the library works inside limits and flags
    when a frame exceeds them.  The C++ nodes act on the flags: a scene with 9 persons (9 detections per camera, up to 19 hypotheses with the
    spurious ones that gross outliers and missed views spawn, 10 tracks) is replayed through contexts that start with room for 3 — a third of what
    the scene needs — and must publish exactly what a replay with ample limits publishes, and what the oracle publishes."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host")], stdout=subprocess.DEVNULL)
    frames, _ = S.generate_sequence(extr_rig, S.SceneConfig(n_persons=9, n_frames=16, seed=31, spacing=0.9, outlier_frac=0.01, p_miss=0.05))
    stream = Hn.stream_from_frames(frames)
    scene, out_small, out_big = str(tmp_path / "scene.bin"), str(tmp_path / "small.bin"), str(tmp_path / "big.bin")
    Hn.write_replay_scene(scene, extr_rig, demo_params, stream)
    exe = os.path.join(ROOT, "host", "se3d_replay")
    small = subprocess.run([exe, scene, out_small], check=True, capture_output=True, text=True, env=dict(os.environ, SE3D_REPLAY_MAX_DET="3", SE3D_REPLAY_MAX_HYP="3"))
    big = subprocess.run([exe, scene, out_big], check=True, capture_output=True, text=True, env=dict(os.environ, SE3D_REPLAY_MAX_DET="32", SE3D_REPLAY_MAX_HYP="128"))
    lim_small = [ln for ln in small.stdout.split("\n") if ln.startswith("Limits:")][0]
    lim_big = [ln for ln in big.stdout.split("\n") if ln.startswith("Limits:")][0]
    assert int(lim_small.split()[2]) >= 2 and int(lim_big.split()[2]) == 0, (lim_small, lim_big)
    a, b = Hn.read_replay_output(out_small), Hn.read_replay_output(out_big)
    assert len(a) == len(b) == len(stream)
    n_pub = 0
    for ra, rb in zip(a, b):
        assert ra is not None and rb is not None
        for la, lb in zip(ra, rb):     # persons_3d, persons3d_fused, persons3d_fused_pred: byte-equal, whatever the limits the run started with
            assert la.persons.tobytes() == lb.persons.tobytes()
        n_pub += len(ra[1].persons)
    assert max(len(r[0].persons) for r in a) >= 9 and n_pub > 0
    # ... and equal to the oracle (which has no limits)
    pp = oracle.PosePrior(demo_params)
    h1 = Hn.Skeleton3DHarness(16, lambda fr: oracle.triangulate(extr_rig, demo_params, fr, capacity=256, hyp_capacity=256))
    h2 = Hn.PosePriorHarness(lambda persons, n_cam, t, fb: pp.step(persons, n_cam, t, fb, capacity=256))
    for k, msgs in enumerate(stream):
        p3d = h1.step(msgs)
        g3d, gfused, _ = a[k]
        assert len(g3d.persons) == len(p3d.persons)
        v = p3d.persons["keypoints"]["score"] > 0
        assert np.array_equal(g3d.persons["keypoints"]["score"] > 0, v)
        ofused, _ = h2.callback(p3d)
        assert len(gfused.persons) == len(ofused.persons)
        if len(ofused.persons):
            assert np.array_equal(gfused.persons["id"], ofused.persons["id"])


def _sharded_replay_equals_single_process(tmp_path, extr_rig, demo_params, ranks, backend):
    """se3d_replay --ranks N: N fresh child processes replay the scene with the hypotheses / detections split over the ranks
    (host/se3d_exchange.cpp: se3d::HypothesisShard on a Collective); every rank's output file must equal the single-process replay's byte for
    byte — stage-1 persons, fused and predicted persons, ids, headers."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host")], stdout=subprocess.DEVNULL)
    frames, _ = S.generate_sequence(extr_rig, S.SceneConfig(n_persons=6, n_frames=14, seed=21, p_miss=0.1, low_score_frac=0.1, outlier_frac=0.02))
    stream = Hn.stream_from_frames(frames)
    scene, out, out_s = str(tmp_path / "scene.bin"), str(tmp_path / "out.bin"), str(tmp_path / "out_sharded.bin")
    Hn.write_replay_scene(scene, extr_rig, demo_params, stream)
    # the same (fixed) limits on both sides, and se3d_refine as per-step launches: the shard calls run the per-step LM kernel, whose sums differ from the
    # fused walk's three-persons-per-wavefront LM in the last bits
    env = dict(os.environ, SE3D_REPLAY_MAX_DET="8", SE3D_REPLAY_MAX_HYP="16", SE3D_REPLAY_REFINE_STEPS="1")
    exe = os.path.join(ROOT, "host", "se3d_replay")
    subprocess.run([exe, scene, out], check=True, env=env, capture_output=True)
    log = subprocess.run([exe, scene, out_s, "--ranks", str(ranks), "--backend", backend], check=True, env=env, capture_output=True, text=True, timeout=300).stdout
    ref = open(out, "rb").read()
    assert sum(len(g[1].persons) for g in Hn.read_replay_output(out) if g is not None) > 10
    for r in range(ranks):
        assert open("%s.rank%d" % (out_s, r), "rb").read() == ref, "rank %d of %d published something else than the single process" % (r, ranks)
    return log


def test_cpp_replay_rccl_world1(tmp_path, extr_rig, demo_params):
    """The RCCL backend of the C++ exchange (librccl.so through dlopen: ncclGetUniqueId / ncclCommInitRank / ncclAllGather on the context's
    stream) with one rank — all this box has; the collective runs and the result equals the un-sharded replay."""
    log = _sharded_replay_equals_single_process(tmp_path, extr_rig, demo_params, 1, "rccl")
    assert "collective: rccl" in log, log


def test_cpp_replay_two_ranks_host_staged(tmp_path, extr_rig, demo_params):
    """Two ranks on the ONE GPU through the test-only host-staged collective (RCCL refuses two ranks on one device): the N > 1 code path of the
    C++ host layer end to end — launcher, rendezvous, shard layout, both all-gathers per frame, replicated association and lifecycle."""
    log = _sharded_replay_equals_single_process(tmp_path, extr_rig, demo_params, 2, "staged")
    assert log.count("host-staged files") == 2, log


@pytest.mark.parametrize("ranks,backend", [(1, "rccl"), (2, "staged")])
def test_cpp_walk_exchange(tmp_path, extr_rig, demo_params, ranks, backend):
    """The throughput sharding's exchange from the C++ host layer (se3d::SkeletonExchange): every rank runs stage 1 and the stage-2 walk over the scene as one
    resident batch, packs the published persons densely on the device and all-gathers persons and offsets on the context's stream; se3d_replay --walk-exchange
    checks inside every rank that its own block equals se3d_fetch_stage2_walk slot by slot and that every rank's block is the same (same scene)."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "host")], stdout=subprocess.DEVNULL)
    frames, _ = S.generate_sequence(extr_rig, S.SceneConfig(n_persons=5, n_frames=14, seed=8, p_miss=0.05))
    scene = str(tmp_path / "scene.bin")
    Hn.write_replay_scene(scene, extr_rig, demo_params, Hn.stream_from_frames(frames))
    env = dict(os.environ, SE3D_REPLAY_MAX_DET="8", SE3D_REPLAY_MAX_HYP="16")
    r = subprocess.run([os.path.join(ROOT, "host", "se3d_replay"), scene, str(tmp_path / "unused.bin"), "--ranks", str(ranks), "--backend", backend, "--walk-exchange"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
    ok = [ln for ln in r.stdout.split("\n") if ln.startswith("walk exchange ok")]
    assert len(ok) == ranks, r.stdout
    assert all(int(ln.split(",")[1].split()[0]) > 10 for ln in ok), ok   # published persons (the tracks publish from the 11th observation on)
