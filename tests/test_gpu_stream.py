"""GPU (-m gpu): the streaming ingest path (se3d_stream_submit / se3d_stream_collect: ragged pinned H2D, expansion on the device, stage 1,
stage-2 walk, device compaction of the published persons, D2H, two batches in flight) returns the same bytes as the resident path
(se3d_stage_frames / se3d_run_stage1 / se3d_run_stage2_walk / se3d_fetch_stage2_walk) on the same frames and stamps."""
import numpy as np
import pytest

from smartedgesensor3dhumanpose_amd import scene as S

pytestmark = pytest.mark.gpu


def _node(rig, **kw):
    from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
    return Skeleton3D(rig, max_epi_dist=0.045, **kw)


# 0: device buffers + D2H copies (default), 1: the compaction kernel writes the pinned host buffers, "toggle": the option changes between batches IN
# FLIGHT (se3d_set_options between two submits: a batch keeps the form it was submitted with — ADVICE r5)
@pytest.mark.parametrize("stream_out", [0, 1, "toggle"])
def test_streaming_equals_resident(extr_rig, stream_out):
    toggle = stream_out == "toggle"
    stream_out = 0 if toggle else stream_out
    S_, T_, n_batches = 6, 16, 5
    seqs = [S.generate_sequence(extr_rig, S.SceneConfig(n_persons=3 + s % 4, n_frames=T_ * n_batches, seed=900 + s, p_miss=0.05 * (s % 3)))[0] for s in range(S_)]
    res, stream = _node(extr_rig, max_frames=S_ * T_, max_det_per_cam=8, max_hyp=12), _node(extr_rig, max_frames=S_ * T_, max_det_per_cam=8, max_hyp=12, options=dict(stream_out=stream_out))
    res.refine_reset()
    stream.refine_reset()
    slots = [k * S_ for k in range(T_)]
    batches, stamps = [], []
    for b in range(n_batches):
        frames = [seqs[s][b * T_ + k] for k in range(T_) for s in range(S_)]
        batches.append(stream.stream_prepare(frames))
        stamps.append(np.array([[seqs[s][b * T_ + k].stamp for s in range(S_)] for k in range(T_)]))
    # resident reference, batch by batch
    expect = []
    for b in range(n_batches):
        res.stage(batches[b][1])
        res.run_stage1()
        res.run_stage2_walk(stamps[b], slots)
        got = [r for k in range(T_) for r in res.fetch_stage2_walk(k * S_, S_)]
        expect.append(got)
    # streaming: two batches in flight
    out = []
    stream.stream_submit(batches[0], stamps[0], slots)
    for b in range(1, n_batches):
        if toggle:
            stream.set_options(stream_out=b % 2)
        stream.stream_submit(batches[b], stamps[b], slots)
        out.append(stream.stream_collect())
    out.append(stream.stream_collect())
    published = 0
    for b in range(n_batches):
        off, fused, pred = out[b]
        assert len(off) == S_ * T_ + 1 and off[-1] == len(fused) == len(pred)
        for i, r in enumerate(expect[b]):
            n = r.n_persons
            assert off[i + 1] - off[i] == n
            assert fused[off[i]:off[i + 1]].tobytes() == r.fused[:n].tobytes()
            assert pred[off[i]:off[i + 1]].tobytes() == r.fused_pred[:n].tobytes()
            published += n
    assert published > S_ * T_ * (n_batches - 1) * 3
    from smartedgesensor3dhumanpose_amd.skeleton_3d import Se3dError
    stream.stream_submit(batches[0], stamps[0], slots)
    stream.stream_submit(batches[1], stamps[1], slots)
    with pytest.raises(Se3dError):
        stream.stream_submit(batches[2], stamps[2], slots)   # only two batches may be in flight
    stream.stream_collect()
    stream.stream_collect()
    with pytest.raises(Se3dError):
        stream.stream_collect()
    res.close()
    stream.close()
