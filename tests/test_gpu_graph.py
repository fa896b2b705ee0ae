"""GPU (-m gpu): the one-frame calls of a max_frames = 1 context (the reference's operating point: one synchronised frame per callback,
S3D:1069 / PP:505) replay ONE hipGraph each — upload, kernel chain, download (round 4).  The replay must give the bytes of the directly
launched calls (options.graph = -1), survive a change of a kernel-form option (re-capture) and the kernel timing being switched on and off."""
import os
import sys

import numpy as np
import pytest

from smartedgesensor3dhumanpose_amd import scene as S

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import branch_scenes  # noqa: E402

pytestmark = pytest.mark.gpu


def _run(node, frames):
    node.refine_reset()
    out = []
    for fr in frames:
        r = node.triangulate_persons([fr])[0]
        f = node.refine(0, r.published(), fr.stamp, fr.fb_delay)
        out.append((r.flags, r.n_hyp, r.assoc().tobytes(), r.published().tobytes(), f.n_persons, int(f.c.n_tracks),
                    f.fused[: f.n_persons].tobytes(), f.fused_pred[: f.n_persons].tobytes(), f.lm_iterations[: f.n_persons].tobytes()))
    return out


def test_graph_replay_equals_direct_launches(extr_rig):
    from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
    frames = S.generate_sequence(extr_rig, S.SceneConfig(n_persons=5, n_frames=14, seed=2, p_miss=0.2))[0] + branch_scenes.outliers(extr_rig, n_frames=3)
    node = Skeleton3D(extr_rig, max_epi_dist=0.045, max_frames=1, max_det_per_cam=8, max_hyp=16, device=0)
    got = _run(node, frames)                       # graph replay (captured by the first call)
    again = _run(node, frames)                     # the same graphs, replayed
    node.set_options(graph=-1)
    direct = _run(node, frames)
    node.set_options(graph=0)
    assert got == again == direct
    assert sum(g[4] for g in got) > 0              # persons were published and refined
    # a kernel-form option changes the kernels of the chain: the graph is captured again and still gives the same results
    node.set_options(post_form="block", refine_steps=1)
    switched = _run(node, frames)
    node.set_options(post_form="auto", refine_steps=0)
    for a, b in zip(got, switched):
        assert a[:6] == b[:6]                      # stage 1 bytewise (post-processing forms agree bytewise), ids / counts equal
    assert _run(node, frames) == got               # ... and back
    # kernel timing on: the calls launch directly (events between the kernels); off again: the graphs are still valid
    node.profile(True)
    timed = _run(node, frames)
    node.profile(False)
    assert timed == got and node.profile_get("sweep")[1] > 0
    assert _run(node, frames) == got
    node.close()
