"""GPU (-m gpu): hypothesis-shard mode (north_star config 5) — W "ranks" emulated by W contexts on one GPU, the all-gather
done by the test (concatenation of the ranks' pack buffers in rank order, exactly what dist.all_gather_into_tensor produces).  Every
rank must end with the single-rank results BIT FOR BIT: stage-1 persons / association, stage-2 fused + predicted persons, ids, LM
iteration counts, track counts — over a multi-frame sequence (track state has to stay identical on all ranks)."""
import numpy as np
import pytest

from smartedgesensor3dhumanpose_amd import rig as R
from smartedgesensor3dhumanpose_amd import scene as S
from smartedgesensor3dhumanpose_amd.skeleton_3d import rig_tensors

pytestmark = pytest.mark.gpu


def _node(rig, **kw):
    from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
    return Skeleton3D(rig, max_epi_dist=0.045, **kw)


def _same_stage1(a, b):
    assert (a.n_persons, a.n_hyp, a.flags) == (b.n_persons, b.n_hyp, b.flags)
    assert np.array_equal(a.assoc(), b.assoc()) and np.array_equal(a.person_hyp[: a.n_persons], b.person_hyp[: b.n_persons])
    assert a.published().tobytes() == b.published().tobytes()


def _same_stage2(a, b):
    assert (a.n_persons, a.c.n_tracks, a.c.pred_delta_t) == (b.n_persons, b.c.n_tracks, b.c.pred_delta_t)
    n = a.n_persons
    assert a.fused[:n].tobytes() == b.fused[:n].tobytes() and a.fused_pred[:n].tobytes() == b.fused_pred[:n].tobytes()
    assert np.array_equal(a.lm_iterations[:n], b.lm_iterations[:n])


@pytest.mark.parametrize("world,rig_name,persons,max_det,max_hyp", [(2, "extr16", 6, 8, 12), (3, "extr16", 7, 8, 16), (4, "ell24", 14, 24, 48), (8, "extr16", 5, 8, 12)])
def test_hypothesis_shard_equals_single_rank(extr_rig, oracle, device_buffer, world, rig_name, persons, max_det, max_hyp):
    if rig_name == "extr16":
        rig = extr_rig
    else:
        rig = R.elliptical_rig(24, seed=3)
        rig.set_tensors(*rig_tensors(rig))   # product tensors (bit-equal to the oracle's: tests/test_abi.py)
    S_, T_ = 3, 14
    extra = dict(p_miss=0.1, low_score_frac=0.1, outlier_frac=0.02)
    seqs = [S.generate_sequence(rig, S.SceneConfig(n_persons=persons - s, n_frames=T_, seed=500 + s, min_coverage=4 if rig.n_cam == 16 else 6, **extra))[0] for s in range(S_)]
    frames = [seqs[s][k] for k in range(T_) for s in range(S_)]
    B = S_ * T_
    kw = dict(max_frames=B, max_det_per_cam=max_det, max_hyp=max_hyp)
    single = _node(rig, **kw)
    single.stage(frames)
    single.run_stage1()
    ref1 = single.fetch_stage1(B)
    single.refine_reset()
    ranks = [_node(rig, **kw) for _ in range(world)]
    for r, nd in enumerate(ranks):
        nd.set_shard(r, world)
        nd.stage(frames)
        nd.refine_reset()
    # ---- stage 1: replicated association, own hypotheses, all-gather, replicated post-processing ----
    nb1 = ranks[0].shard_bytes(1, B)
    packs = [device_buffer(nb1) for _ in ranks]
    gath = [device_buffer(nb1 * world) for _ in ranks]
    for nd, pk in zip(ranks, packs):
        nd.run_stage1_shard_pre(pk.data_ptr())
        nd.sync()
    allp = np.concatenate([pk.numpy(np.uint8) for pk in packs])      # = all_gather_into_tensor
    for nd, g in zip(ranks, gath):
        g.upload(allp)
        nd.run_stage1_shard_post(g.data_ptr())
        nd.sync()
    for nd in ranks:
        for a, b in zip(nd.fetch_stage1(B), ref1):
            _same_stage1(a, b)
    assert sum(r.n_persons for r in ref1) > B * (persons - 3)
    # ---- stage 2, step by step: replicated track association, own detections refined, all-gather of records, replicated lifecycle ----
    nb2 = ranks[0].shard_bytes(2, S_)
    packs2 = [device_buffer(nb2) for _ in ranks]
    gath2 = [device_buffer(nb2 * world) for _ in ranks]
    published = 0
    for k in range(T_):
        t = np.array([seqs[s][k].stamp for s in range(S_)])
        single.run_stage2(t, frame_slot0=k * S_)
        ref2 = single.fetch_stage2(S_)
        for nd, pk in zip(ranks, packs2):
            nd.run_stage2_shard_pre(t, k * S_, pk.data_ptr())
            nd.sync()
        allr = np.concatenate([pk.numpy(np.uint8) for pk in packs2])
        for nd, g in zip(ranks, gath2):
            g.upload(allr)
            nd.run_stage2_shard_post(S_, g.data_ptr())
            nd.sync()
        for nd in ranks:
            for a, b in zip(nd.fetch_stage2(S_), ref2):
                _same_stage2(a, b)
        published += sum(r.n_persons for r in ref2)
    assert published > S_ * (persons - 3)
    for nd in ranks + [single]:
        nd.close()
