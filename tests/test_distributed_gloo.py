"""N>1 path on CPU (gloo, world_size 2): sequence sharding + all-gather of final skeletons reproduce the single-process
result.  The per-rank compute is stood in for by the oracle (test infrastructure); the exchange code is the product's
(smartedgesensor3dhumanpose_amd/shard.py), the same functions bench.py uses over RCCL."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SEQ, MAX_HYP = 5, 16


def _scene(seq):
    from smartedgesensor3dhumanpose_amd import rig as R, scene as S
    from oracle import oracle_py as O
    rig = R.extr_rig_16cam()
    rig.set_tensors(*O.rig_tensors(rig))
    frames, _ = S.generate_sequence(rig, S.SceneConfig(n_persons=2 + seq % 3, n_frames=1, seed=70 + seq))
    return rig, frames[0]


def _compute(seq):
    from smartedgesensor3dhumanpose_amd import person_msgs as pm
    from oracle import oracle_py as O
    rig, fr = _scene(seq)
    return O.triangulate(rig, pm.Params(pm.POSE_SIMPLE, 0, 0.045), fr, capacity=MAX_HYP, hyp_capacity=MAX_HYP).published().copy()


def _worker(rank, world, port, q, form="slots"):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from smartedgesensor3dhumanpose_amd import shard
    mine = shard.rank_sequences(N_SEQ, rank, world)
    n_slots = (N_SEQ + world - 1) // world
    results = [_compute(s) for s in mine] + [np.zeros(0, dtype=_compute(0).dtype)] * (n_slots - len(mine))
    if form == "dense":   # the exchange form of the small workloads: persons back to back + offsets (se3d_pack_stage2_walk_device_dense)
        packed, offs = shard.pack_host_dense(results, 4 * n_slots)
        gp, go = shard.gather_skeletons(dist, packed, offs, MAX_HYP)
        allp = []
        for r in range(world):
            persons, dropped = shard.unpack_dense(gp[r], go[r], n_slots)
            assert dropped == 0
            allp.append(persons)
    else:
        packed, counts = shard.pack_host(results, MAX_HYP)
        gp, gc = shard.gather_skeletons(dist, packed, counts, MAX_HYP)
        allp = shard.unpack_skeletons(gp, gc, MAX_HYP)
    # every rank now holds all skeletons: re-assemble in sequence order
    out = {}
    for r in range(world):
        for slot, s in enumerate(shard.rank_sequences(N_SEQ, r, world)):
            out[s] = allp[r][slot]
    q.put((rank, {s: v.tobytes() for s, v in out.items()}))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,form", [(2, "slots"), (4, "slots"), (2, "dense"), (4, "dense")])
def test_sequence_sharding_and_allgather(world, form):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + 7 * world + (3 if form == "dense" else 0)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, form)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    ref = {s: _compute(s).tobytes() for s in range(N_SEQ)}
    for rank, out in got:
        assert sorted(out) == list(range(N_SEQ))
        for s in range(N_SEQ):
            assert out[s] == ref[s], (rank, s)


def test_rank_sequences_partition():
    from smartedgesensor3dhumanpose_amd import shard
    for world in (1, 2, 4, 8):
        seen = sorted(sum((shard.rank_sequences(13, r, world) for r in range(world)), []))
        assert seen == list(range(13))


# ---- bench.py's own exchange object (shard.SkeletonExchange: double-buffered, one all-gather pair per step) on two gloo ranks ----
N_CTX, SLICE_P, SLICE_C, N_STEPS = 3, 1688 * 5, 5, 5


def _pattern(rank, ci, step):
    rng = np.random.default_rng(1000 * rank + 10 * ci + step)
    return rng.integers(0, 256, SLICE_P, dtype=np.uint8), rng.integers(0, 7, SLICE_C, dtype=np.int32)


def _xworker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from smartedgesensor3dhumanpose_amd import shard
    ex = shard.SkeletonExchange(dist, N_CTX, SLICE_P, SLICE_C, torch.device("cpu"), streams=None)
    ex.timed(True)           # the event stamps exist on CUDA streams only: on gloo the switch must be inert (bench.py calls it on every backend)
    ok = True
    for step in range(N_STEPS):
        for ci in range(N_CTX):
            pk, cn = ex.pack_target(ci)
            p, c = _pattern(rank, ci, step)
            pk.copy_(torch.from_numpy(p))
            cn.copy_(torch.from_numpy(c))
            ex.packed(ci)
        par = ex.exchange()
        assert par == step % 2                      # double-buffered
        ex.finish()
        gp, gc = ex.gathered(par)
        for r in range(world):                      # every rank's gathered buffer == concatenation of all ranks' packs of THIS step
            for ci in range(N_CTX):
                p, c = _pattern(r, ci, step)
                ok = ok and np.array_equal(gp[r, ci].numpy(), p) and np.array_equal(gc[r, ci].numpy(), c)
        ok = ok and ex.verify(par)
    ok = ok and ex.exchange_ms() is None and not ex.timing
    # a step with the all-gather skipped (bench.py's value_no_exchange pass) writes the SAME pack buffer again and leaves the gathered ones alone
    before = ex.gath_p[par].clone()
    pk, cn = ex.pack_target(0)
    ok = ok and pk.data_ptr() == ex.packs[ex.parity].data_ptr()
    pk.zero_()
    ex.packed(0)
    ok = ok and torch.equal(before, ex.gath_p[par]) and ex.n_exchanges == N_STEPS
    # a corrupted slice must be detected by verify() on every rank
    ex.gath_p[par][-1] ^= 0xFF
    bad = not ex.verify(par)
    q.put((rank, ok, bad, ex.n_exchanges))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_skeleton_exchange_gathers_all_packs(world):
    """world 4 and 8 (round 3): the layout [world][n_ctx][slice] with N_CTX = 3 contexts per rank, every rank checks every other rank's slices."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + 7 * world
    procs = [ctx.Process(target=_xworker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, bad, n in got:
        assert ok, "rank %d: gathered buffer differs from the concatenation of the packs" % rank
        assert bad, "rank %d: verify() did not notice a corrupted slice" % rank
        assert n == N_STEPS


def test_bench_launcher_builds_a_torchrun_command(monkeypatch):
    """bench.py --gpus N without RANK starts N ranks itself: the launcher must be invoked before anything touches the GPU."""
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("RANK", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                   # the children's return code is forwarded
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert "torch.cuda" not in sys.modules or not sys.modules["torch"].cuda.is_initialized()


# ---- hypothesis shard of one stream (north_star config 5): layout + collective on two gloo ranks, compute = the oracle ----
def _hworker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from smartedgesensor3dhumanpose_amd import person_msgs as pm, shard
    from oracle import oracle_py as O
    params = pm.Params(pm.POSE_SIMPLE, 0, 0.045)
    frames = [_scene(s)[1] for s in range(3)]
    rig = _scene(0)[0]
    H = 10   # not a multiple of the world size: the last block is short
    blocks = np.zeros((len(frames), H), pm.person3d_dt)
    for k, fr in enumerate(frames):
        a = O.triangulate(rig, params, fr, capacity=H, hyp_capacity=H).assoc()      # the association is replicated: every rank computes it
        table = np.full((H, rig.n_cam), -1, np.int32)
        table[: len(a)] = a
        own = table.copy()
        own[[h for h in range(H) if h % world != rank]] = -1                         # ... and triangulates only its own hypotheses
        blocks[k] = O.triangulate_assoc(rig, params, fr, own)
    mine = shard.pack_own(blocks, rank, world)
    flat = torch.from_numpy(mine.view(np.uint8).reshape(-1).copy())
    gath = torch.empty(world * flat.numel(), dtype=torch.uint8)
    dist.all_gather_into_tensor(gath, flat)
    allh = shard.unpack_all(gath.numpy().view(pm.person3d_dt).reshape(world, len(frames), shard.shard_slots(H, world)), H)
    q.put((rank, allh.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_hypothesis_shard_layout_equals_single_rank(world):
    """H = 10 hypotheses over 2, 4 and 8 ranks: H % world != 0 for 4 and 8 (short last blocks, ranks 2.. of world 8 own one hypothesis or none)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + 7 * world
    procs = [ctx.Process(target=_hworker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    from smartedgesensor3dhumanpose_amd import person_msgs as pm
    from oracle import oracle_py as O
    params = pm.Params(pm.POSE_SIMPLE, 0, 0.045)
    rig = _scene(0)[0]
    ref = np.zeros((3, 10), pm.person3d_dt)
    for k in range(3):
        fr = _scene(k)[1]
        a = O.triangulate(rig, params, fr, capacity=10, hyp_capacity=10).assoc()
        table = np.full((10, rig.n_cam), -1, np.int32)
        table[: len(a)] = a
        ref[k] = O.triangulate_assoc(rig, params, fr, table)
    assert (ref["keypoints"]["score"] > 0).any()
    for rank, blob in got:
        assert blob == ref.tobytes(), "rank %d: gathered hypotheses differ from the single-rank result" % rank


def test_shard_pack_unpack_roundtrip():
    from smartedgesensor3dhumanpose_amd import shard
    rng = np.random.default_rng(0)
    for world in (1, 2, 3, 8):
        for H in (1, 7, 8, 12):
            x = rng.integers(0, 1000, (5, H, 3))
            g = np.stack([shard.pack_own(x, r, world) for r in range(world)])
            assert g.shape[2] == shard.shard_slots(H, world)
            assert np.array_equal(shard.unpack_all(g, H), x)


def test_bench_guard_reports_exceptions_and_deadlines():
    """bench.guarded (the wrapper of the hypothesis-shard block, which runs last so that a failure of that mode cannot lose the scaling
    line): result, exception text, deadline."""
    import time
    sys.path.insert(0, ROOT)
    import bench
    assert bench.guarded(lambda: 41 + 1, 5.0, "block") == (42, None, False)
    res, err, hung = bench.guarded(lambda: 1 // 0, 5.0, "block")
    assert res is None and "ZeroDivisionError" in err and not hung
    res, err, hung = bench.guarded(lambda: time.sleep(1.5), 0.1, "block")
    assert res is None and "did not finish" in err and hung
