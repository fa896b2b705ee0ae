"""GPU (-m gpu): bench.py as the driver runs it — under the torch.distributed launcher with one rank (the RCCL exchange path with
world_size 1: pack on the context streams, event-ordered all-gather, gathered buffer verified against the packs), the config-2 line, and —
round 5 — the N > 1 path with TWO ranks on the one GPU of the box through a test-only exchange backend."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def test_bench_exchange_under_launcher_world1():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29631",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "16cam6p", "--seqs", "32", "--contexts", "2", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = _json_line(p.stdout)
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert "verified: True" in line["config"]["sharding"]
    assert line["roofline"]["frac"] > 0 and line["roofline"]["valu"]["achieved"] > 0
    assert line["exchange_ms_per_step"] >= 0 and line["value_no_exchange"] > 0 and line["collective"]["world_size"] == 1


def test_bench_256cam128p_under_launcher_with_hypothesis_shard_and_exchange_diagnosis():
    """The code path of the driver's multi-GPU run at north_star's configuration, with one rank: 256cam128p under the launcher (RCCL exchange
    of the slot form, verified), the exchange diagnosis keys of round 4, and the hypothesis-shard block behind its deadline."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29633",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--workload", "256cam128p", "--contexts", "1", "--seqs", "1", "--tsteps", "4", "--steps", "2", "--warmup", "1",
           "--hyp-shard", "--hyp-deadline-s", "600", "--no-cpu-baseline"]
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    line = _json_line(p.stdout)
    assert line["n_gpus"] == 1 and line["value"] > 0 and "verified: True" in line["config"]["sharding"]
    assert line["exchange_ms_per_step"] is not None and line["exchange_ms_per_step"] >= 0 and line["value_no_exchange"] > 0
    assert line["collective"]["world_size"] == 1 and line["collective"]["backend"] == "nccl"
    hs = line["hypothesis_shard"]
    assert "error" not in hs, hs
    assert hs["verified_equal_to_single_rank"] is True and hs["skeletons_per_s"] > 0


def test_bench_config2_line():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--stage", "dlt", "--seqs", "256", "--steps", "3", "--warmup", "1"]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    line = _json_line(p.stdout)
    assert line["value"] > 0 and line["latency_B1"]["device_ms"] > 0 and "config 2" in line["config"]["workload"]


def test_bench_parity_and_latency_blocks():
    """The untimed blocks of the default driver line (round 3): MPJPE / association exactness against the oracle on the bench's own scenes, and
    the one-frame-per-call latency of se3d_triangulate + se3d_refine."""
    sys.path.insert(0, ROOT)
    import bench
    w = bench.WORKLOADS["16cam6p"]
    rig = bench.make_rig(w)
    seqs = bench.make_sequences(None, w, 1, 2, 0, 16)[0]
    p = bench.bench_parity("16cam6p", rig, seqs, 2)
    assert p["frames"] == 32 and p["assoc_exact_rate"] == 1.0 and p["track_ids_exact_rate"] == 1.0 and p["lm_iterations_equal_rate"] == 1.0
    assert p["mpjpe_max_mm"] < 1.0 and p["refined_max_mm"] < 1.0 and p["refined_skeletons"] > 0
    lat = bench.bench_latency_b1(rig, seqs[0], "16cam6p", frames_timed=30)
    assert 0 < lat["device_ms"]["frame"] < lat["host_call_ms"]["frame"] < 50 and lat["refined_skeletons_per_frame"] > 0


def _launch_two_ranks_on_one_gpu(port, extra):
    """bench.py --gpus 2 with BOTH ranks on device 0: RCCL refuses two ranks on one device, so the exchange goes through the test-only backend
    (--exchange-backend gloo-staged: shard.HostStagedDist, device buffers staged through host memory, gloo between the processes).  Fresh
    child processes from the launcher before anything touches the GPU, as in the driver's multi-GPU run."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--exchange-backend", "gloo-staged", "--no-cpu-baseline"] + extra
    p = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-3000:])
    return _json_line(p.stdout)


def test_bench_two_ranks_16cam6p_rank_partition_exchange_and_timing_reduce():
    """The N > 1 code path of the headline workload with world = 2 (never executed on a GPU before round 5): every rank generates and owns its
    own sequences (seed = 1000 rank + ...), packs its contexts' walk results in the dense form, the all-gather pair is enqueued behind the pack
    events, every rank verifies the gathered buffer against checksums from the senders (exchange_verified), the accounting step asserts that no
    remote frame was truncated, the timings are max-reduced over the ranks and `value` counts the skeletons of BOTH ranks."""
    line = _launch_two_ranks_on_one_gpu(29641, ["--workload", "16cam6p", "--seqs", "32", "--contexts", "2", "--steps", "3", "--warmup", "1"])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["scaling"] == "weak"
    assert "verified: True" in line["config"]["sharding"]
    assert line["collective"]["world_size"] == 2 and "gloo" in line["collective"]["backend"]
    assert line["exchange_ms_per_step"] is not None and line["value_no_exchange"] > 0
    per_rank = line["config"]["frames_per_step"]   # frames per step and GPU
    assert per_rank == 2 * 32 * 16
    # both ranks' skeletons are in the count: ~6 published persons per frame (a few tracks of a pass are not yet published)
    skel = line["value"] * line["ms_per_step"] * 1e-3
    assert 2 * per_rank * 5.0 < skel <= 2 * per_rank * 6.0, skel


def test_bench_two_ranks_256cam128p_with_hypothesis_shard():
    """north_star's configuration with world = 2 on one GPU, one small context per rank (1 sequence x 4 frames), incl. the hypothesis-shard block:
    both ranks stage the SAME stream, triangulate / refine the hypotheses with index % 2 == rank, all-gather the blocks and must end with walks
    bit-identical to the un-sharded walk of the same context."""
    line = _launch_two_ranks_on_one_gpu(29643, ["--workload", "256cam128p", "--contexts", "1", "--seqs", "1", "--tsteps", "4", "--steps", "2", "--warmup", "1",
                                                "--hyp-shard", "--hyp-deadline-s", "900"])
    assert line["n_gpus"] == 2 and line["value"] > 0 and "verified: True" in line["config"]["sharding"]
    assert line["collective"]["world_size"] == 2
    hs = line["hypothesis_shard"]
    assert "error" not in hs, hs
    assert hs["n_gpus"] == 2 and hs["verified_equal_to_single_rank"] is True and hs["skeletons_per_s"] > 0
