#!/usr/bin/env python3
"""bench.py — 3D skeletons/sec of the MI355X-native skeleton_3d (+ pose_prior) hot path on synthetic scenes.

One "step" = one pass of the hot path over one batch of B synthetic frames that are already resident in HBM.
`value` = published 3D skeletons (person-frames) per second over all ranks.  See DESIGN.md §measurement.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 16cam6p|64cam32p|256cam128p] [--frames B]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (n_cam, n_persons, default frames/batch, max_det_per_cam, max_hyp, bytes/skeleton stage 1)
    "16cam6p": dict(n_cam=16, n_persons=6, frames=2048, max_det=8, max_hyp=16, bytes_per_skel=8972),
    "64cam32p": dict(n_cam=64, n_persons=32, frames=64, max_det=40, max_hyp=96, bytes_per_skel=30072),
    "256cam128p": dict(n_cam=256, n_persons=128, frames=4, max_det=160, max_hyp=256, bytes_per_skel=115320),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def make_rig(w):
    from smartedgesensor3dhumanpose_amd import rig as R
    from smartedgesensor3dhumanpose_amd.skeleton_3d import rig_tensors
    rig = R.extr_rig_16cam() if w["n_cam"] == 16 else R.elliptical_rig(w["n_cam"], seed=0)
    rig.set_tensors(*rig_tensors(rig))
    return rig


def make_frames(rig, w, n_frames, seed0):
    """n_frames synthetic frames: distinct seeded 16-frame sequences, repeated to fill the batch."""
    from smartedgesensor3dhumanpose_amd import scene as S
    n_unique = min(n_frames, 64)
    seqs = []
    f = 0
    while f < n_unique:
        cfg = S.SceneConfig(n_persons=w["n_persons"], n_frames=min(16, n_unique - f), seed=seed0 + len(seqs),
                            min_coverage=4 if w["n_cam"] == 16 else 6)
        fr, _ = S.generate_sequence(rig, cfg)
        seqs += fr
        f += len(fr)
    return [seqs[i % len(seqs)] for i in range(n_frames)]


def cpu_baseline(rig, frames, params, budget_s=12.0):
    """Oracle (CPU restatement of the reference, 'port') timed on this box's host cores on a bounded sample."""
    from oracle import oracle_py as O
    ncores = os.cpu_count() or 1
    out = {}
    for label, thr in (("1t", 1), ("omp", ncores)):
        t0 = time.perf_counter()
        n_sk, n_fr = 0, 0
        while time.perf_counter() - t0 < budget_s / 2 and n_fr < len(frames):
            r = O.triangulate(rig, params, frames[n_fr], capacity=512, hyp_capacity=512, omp_threads=thr)
            n_sk += r.n_persons
            n_fr += 1
        dt = time.perf_counter() - t0
        out[label] = (n_sk / dt, n_fr, dt)
    best = max(out.values(), key=lambda v: v[0])
    return {"value": round(best[0], 2), "unit": "skeletons/s", "cores": ncores if best is out["omp"] else 1, "kind": "port",
            "sample": "%d frames of the same workload, stage 1 (single-thread %.1f skel/s over %d frames; OpenMP x%d over hypotheses like S3D:676: %.1f skel/s over %d frames)"
                      % (best[1], out["1t"][0], out["1t"][1], ncores, out["omp"][0], out["omp"][1])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="16cam6p", choices=sorted(WORKLOADS))
    ap.add_argument("--frames", type=int, default=0, help="frames per batch per GPU (default: workload preset)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        dist = None
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")

    from smartedgesensor3dhumanpose_amd import person_msgs as pm
    from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
    w = WORKLOADS[args.workload]
    B = args.frames or w["frames"]
    rig = make_rig(w)
    frames = make_frames(rig, w, B, seed0=1000 * rank)  # every rank works on its own frames (weak scaling)
    node = Skeleton3D(rig, max_epi_dist=0.045, max_frames=B, max_det_per_cam=w["max_det"], max_hyp=w["max_hyp"], device=local_rank)
    node.stage(frames)  # inputs resident in HBM before the timed region

    # exchange buffers for the all-gather of final skeletons (every rank ends up with all skeletons)
    H = node.limits.max_hyp
    pack = torch.empty(B * H * pm.person3d_dt.itemsize, dtype=torch.uint8, device="cuda")
    counts = torch.empty(B, dtype=torch.int32, device="cuda")
    if dist is not None:
        gather_p = torch.empty(world * pack.numel(), dtype=torch.uint8, device="cuda")
        gather_c = torch.empty(world * B, dtype=torch.int32, device="cuda")

    def step():
        node.run_stage1()
        if dist is not None:
            node.pack_stage1_device(pack.data_ptr(), counts.data_ptr())  # syncs the context stream
            dist.all_gather_into_tensor(gather_p, pack)
            dist.all_gather_into_tensor(gather_c, counts)

    for _ in range(args.warmup):
        step()
    node.sync()
    node.profile(True)
    node.profile_reset()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    node.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    res = node.fetch_stage1(B)
    n_skel_local = sum(r.n_persons for r in res)
    n_skel = n_skel_local
    if dist is not None:
        tot = torch.tensor([n_skel_local], dtype=torch.int64, device="cuda")
        dist.all_reduce(tot)
        n_skel = int(tot.item())

    if rank == 0:
        prof = {k: node.profile_get(k) for k in ("normalize", "epipolar_pairs", "sweep", "triangulate", "postprocess")}
        dom = max(prof, key=lambda k: prof[k][0])
        ms_dom = prof[dom][0] / max(prof[dom][1], 1)
        alg_bytes = w["bytes_per_skel"] * n_skel_local  # algorithmic bytes one launch (= one batch) processes
        achieved = alg_bytes / (ms_dom * 1e-3) / 1e9 if ms_dom > 0 else 0.0
        line = {
            "metric": "3D skeletons/sec (triangulated, with UT covariance) on synthetic multi-view scenes",
            "value": round(n_skel * args.steps / dt, 2), "unit": "skeletons/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 geometry / f64 accumulation+eigen", "data": "synthetic",
            "config": {"workload": "%s stage1 (association + DLT triangulation + UT covariance + merge), %d frames/batch/GPU, %d skeletons/batch/GPU"
                                   % (args.workload, B, n_skel_local), "n_cam": w["n_cam"], "n_persons": w["n_persons"], "n_joints": 17,
                       "frames_per_batch": B, "sharding": "frames (independent sequences per rank) + all-gather of final skeletons" if world > 1 else "single GPU"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None,
                         "avg_launch_ms": round(ms_dom, 5),
                         "note": "path is VALU/latency-bound, not HBM-bound (SURVEY 8d): algorithmic bytes = %d B/skeleton" % w["bytes_per_skel"],
                         "kernel_ms_per_step": {k: round(v[0] / max(v[1], 1), 5) for k, v in prof.items()}},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(rig, frames, node.params)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    node.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
