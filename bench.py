#!/usr/bin/env python3
"""bench.py — 3D skeletons/sec of the MI355X-native skeleton_3d + pose_prior hot path on synthetic scenes.

One "step" = one pass of the hot path over one batch of synthetic frames that are already resident in HBM:
  stage 1 (association + DLT triangulation + UT covariance + merge) over all B = S x T frames, then
  stage 2 (track association + LM refinement + marginals + prediction) walking the T time steps of the S independent
  sequences.  Time steps are walked forward on even steps and backward on odd steps so that person motion stays
  continuous, tracks persist and (after warm-up) every person-frame is published (num_obs > 10, PP:845).
`value` = published refined 3D skeletons (person-frames) per second over all ranks.  See DESIGN.md §measurement.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 16cam6p|64cam32p|256cam128p] [--seqs S] [--stage 1|12]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# Each context owns one HIP stream; the ROCm runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T_STEPS = 16  # frames per synthetic sequence (>= 16 frames per sequence)
WORKLOADS = {
    # bytes/skeleton: SURVEY §8d algorithmic bytes (stage 1) ; stage 2 adds 8.9 KB per skeleton
    "16cam6p": dict(n_cam=16, n_persons=6, seqs=512, contexts=12, max_det=8, max_hyp=12, bytes_s1=8972, bytes_s2=8900),
    "64cam32p": dict(n_cam=64, n_persons=32, seqs=32, contexts=3, max_det=40, max_hyp=96, bytes_s1=30072, bytes_s2=8900),
    "256cam128p": dict(n_cam=256, n_persons=128, seqs=1, contexts=3, max_det=128, max_hyp=256, bytes_s1=115320, bytes_s2=8900),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def make_rig(w):
    from smartedgesensor3dhumanpose_amd import rig as R
    from smartedgesensor3dhumanpose_amd.skeleton_3d import rig_tensors
    rig = R.extr_rig_16cam() if w["n_cam"] == 16 else R.elliptical_rig(w["n_cam"], seed=0)
    rig.set_tensors(*rig_tensors(rig))
    return rig


def make_sequences(rig, w, n_seq, seed0, t_steps):
    """n_seq synthetic sequences of t_steps frames (up to 32 distinct seeded sequences per context, repeated)."""
    from smartedgesensor3dhumanpose_amd import scene as S
    uniq = []
    for i in range(min(n_seq, 32)):
        cfg = S.SceneConfig(n_persons=w["n_persons"], n_frames=t_steps, seed=seed0 + i, min_coverage=4 if w["n_cam"] == 16 else 6)
        uniq.append(S.generate_sequence(rig, cfg)[0])
    return [uniq[s % len(uniq)] for s in range(n_seq)]


def cpu_baseline(rig, seq, params, with_stage2, budget_s=12.0):
    """Oracle (CPU restatement of the reference, kind 'port') timed on this box's host cores on a bounded sample."""
    from oracle import oracle_py as O
    ncores = os.cpu_count() or 1
    out = {}
    for label, thr in (("1t", 1), ("omp", ncores)):
        pp = O.PosePrior(params) if with_stage2 else None
        t0 = time.perf_counter()
        n_sk, n_fr = 0, 0
        while time.perf_counter() - t0 < budget_s / 2:
            fr = seq[n_fr % len(seq)]
            r = O.triangulate(rig, params, fr, capacity=512, hyp_capacity=512, omp_threads=thr)
            if pp is not None:
                pp.step(r.published(), rig.n_cam, 100.0 + n_fr / 30.0, fr.fb_delay, capacity=512, omp_threads=thr)
            n_sk += r.n_persons
            n_fr += 1
        dt = time.perf_counter() - t0
        out[label] = (n_sk / dt, n_fr, dt)
    best = "omp" if out["omp"][0] > out["1t"][0] else "1t"
    return {"value": round(out[best][0], 2), "unit": "skeletons/s", "cores": ncores if best == "omp" else 1, "kind": "port",
            "sample": "%d frames of one sequence of the same workload, stage %s; single thread %.1f skel/s (%d frames), OpenMP x%d over "
                      "hypotheses/persons as in S3D:676 / PP:582: %.1f skel/s (%d frames)"
                      % (out[best][1], "1+2" if with_stage2 else "1", out["1t"][0], out["1t"][1], ncores, out["omp"][0], out["omp"][1])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="16cam6p", choices=sorted(WORKLOADS))
    ap.add_argument("--seqs", type=int, default=0, help="independent sequences per GPU (default: workload preset)")
    ap.add_argument("--tsteps", type=int, default=T_STEPS)
    ap.add_argument("--stage", default="12", choices=["1", "12"], help="1: stage 1 only; 12: stage 1 + stage 2")
    ap.add_argument("--contexts", type=int, default=0, help="independent se3d contexts (HIP streams) per GPU, each with --seqs sequences; "
                    "their launches interleave so latency-bound phases of one overlap throughput-bound kernels of the other (default: preset)")
    ap.add_argument("--walk", default="fused", choices=["fused", "steps"], help="stage 2 as one multi-step launch per context and step (default) "
                    "or as T per-step calls (A/B)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run (also with one rank: exercises the exchange path)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")

    from smartedgesensor3dhumanpose_amd import person_msgs as pm
    from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
    w = WORKLOADS[args.workload]
    S_, T_ = args.seqs or w["seqs"], args.tsteps
    B = S_ * T_
    stage2 = args.stage == "12"
    rig = make_rig(w)
    NC = args.contexts or w["contexts"]
    seed_rank = int(os.environ.get("SE3D_BENCH_SEED_RANK", rank))  # diagnostic: run another rank's synthetic scenes on this GPU (tools/check_rank_scenes.sh)
    nodes, all_seqs = [], []
    for ci in range(NC):  # every rank (and every context) owns its own sequences: weak scaling, no data-path collective
        seqs = make_sequences(rig, w, S_, seed0=1000 * seed_rank + 100 * ci, t_steps=T_)
        frames = [seqs[s][k] for k in range(T_) for s in range(S_)]   # frame slot = k * S + s
        nd_ = Skeleton3D(rig, max_epi_dist=0.045, max_frames=B, max_det_per_cam=w["max_det"], max_hyp=w["max_hyp"], device=local_rank)
        nd_.stage(frames)  # inputs resident in HBM before the timed region
        nodes.append(nd_)
        all_seqs.append(seqs)
    node, seqs = nodes[0], all_seqs[0]
    dt_frame = 1.0 / 30.0

    H = node.limits.max_hyp
    psz = pm.person3d_dt.itemsize
    if dist is not None:
        # Exchange: ONE RCCL all-gather of the final skeletons per step so that every rank holds the skeletons of all ranks
        # ([contexts][B][n_persons + 2] person slots + [contexts][B] counts, frame-slot order).  Stream-ordered, no host
        # synchronisation: every context packs its slice asynchronously on its own HIP stream (handed to torch as an ExternalStream) and
        # records an event; a dedicated exchange stream waits for the events of all contexts and enqueues the collective, which runs on
        # RCCL's stream while the contexts compute their next step; the pack buffers are double-buffered and a slice is rewritten only
        # after the collective that read it two steps earlier has completed (device-side wait on the context stream).
        n_slots = B
        XH = min(H, w["n_persons"] + 2) if stage2 and args.walk == "fused" else H   # person slots per frame carried by the exchange
        streams = [torch.cuda.ExternalStream(nd_.stream_handle()) for nd_ in nodes]
        xstream = torch.cuda.Stream()
        slice_p, slice_c = n_slots * XH * psz, n_slots
        packs = [torch.zeros(NC * slice_p, dtype=torch.uint8, device="cuda") for _ in range(2)]
        cnts = [torch.zeros(NC * slice_c, dtype=torch.int32, device="cuda") for _ in range(2)]
        gath_p = torch.empty(world * NC * slice_p, dtype=torch.uint8, device="cuda")
        gath_c = torch.empty(world * NC * slice_c, dtype=torch.int32, device="cuda")
        works = [None, None]
        packed_ev = [None] * NC
        xstate = {"parity": 0}

    def pack_target(ci):
        p = xstate["parity"]
        if works[p] is not None:
            with torch.cuda.stream(streams[ci]):
                for w_ in works[p]:
                    w_.wait()
        return packs[p][ci * slice_p:(ci + 1) * slice_p], cnts[p][ci * slice_c:(ci + 1) * slice_c]

    def packed(ci):
        packed_ev[ci] = streams[ci].record_event()

    def exchange():
        p = xstate["parity"]
        with torch.cuda.stream(xstream):
            for ev in packed_ev:
                xstream.wait_event(ev)
            works[p] = (dist.all_gather_into_tensor(gath_p, packs[p], async_op=True),
                        dist.all_gather_into_tensor(gath_c, cnts[p], async_op=True))
        xstate["parity"] ^= 1

    # One step = per context one stage-1 pass over its B resident frames and one stage-2 walk over the T time steps of its S
    # sequences (se3d_run_stage2_walk: one launch, a workgroup per sequence; forward on even cycles, backward on odd ones so that
    # motion stays continuous and tracks persist).  With --walk steps the walk is issued as T per-step calls (se3d_run_stage2).
    # Contexts are phase-shifted (odd contexts issue walk-then-stage-1) so that the throughput-bound stage-1 kernels of one context
    # overlap the latency-bound stage-2 walk of another instead of all contexts running the same phase in lockstep.  Stage-1
    # outputs of a frame slot are a pure function of the resident inputs, so a walk always consumes complete stage-1 results.
    walks = [{"node": nd_, "ci": ci, "cycle": 0, "t": 100.0} for ci, nd_ in enumerate(nodes)]

    def issue_stage1(wk):
        wk["node"].run_stage1()
        if not stage2 and dist is not None:
            pk, cn = pack_target(wk["ci"])
            wk["node"].pack_stage1_device(pk.data_ptr(), cn.data_ptr())
            packed(wk["ci"])

    def issue_walk(wk, collect):
        nd_, ci = wk["node"], wk["ci"]
        ks = list(range(T_)) if wk["cycle"] % 2 == 0 else list(range(T_ - 1, -1, -1))
        ts = wk["t"] + dt_frame * np.arange(1, T_ + 1)
        wk["t"] = float(ts[-1])
        wk["cycle"] += 1
        if args.walk == "fused":
            nd_.run_stage2_walk(np.repeat(ts[:, None], S_, 1), [k * S_ for k in ks])
            if dist is not None:  # async D2D of all time steps' fused skeletons (frame-slot order) into the exchange buffer
                pk, cn = pack_target(ci)
                nd_.pack_stage2_walk_device_slots(B, XH, pk.data_ptr(), cn.data_ptr())
                packed(ci)
            if collect is not None:
                per_frame = [r.n_persons for k in ks for r in nd_.fetch_stage2_walk(k * S_, S_)]
                assert dist is None or max(per_frame) <= XH, "exchange slots per frame too small: skeletons would be truncated"
                max_pub[0] = max(max_pub[0], max(per_frame))
                collect.append(sum(per_frame))
        else:
            pk, cn = pack_target(ci) if dist is not None else (None, None)
            for k, t in zip(ks, ts):
                nd_.run_stage2(np.full(S_, t), frame_slot0=k * S_)
                if dist is not None:
                    nd_.pack_stage2_device(S_, pk.data_ptr() + k * S_ * H * psz, cn.data_ptr() + 4 * k * S_)
                if collect is not None:
                    collect.append(sum(r.n_persons for r in nd_.fetch_stage2(S_)))
            if dist is not None:
                packed(ci)

    max_pub = [0]   # most skeletons any frame published (accounting pass): the exchange carries n_persons + 2 slots per frame
    def step(collect=None):
        if not stage2:
            for wk in walks:
                issue_stage1(wk)
        else:
            for wk in walks:   # phase A
                issue_stage1(wk) if wk["ci"] % 2 == 0 else issue_walk(wk, collect)
            for wk in walks:   # phase B
                issue_walk(wk, collect) if wk["ci"] % 2 == 0 else issue_stage1(wk)
        if dist is not None:
            exchange()

    for nd_ in nodes:
        nd_.run_stage1()   # phase-shifted contexts start inside the stage-2 walk: their frames need one completed stage-1 pass
    for nd_ in nodes:
        nd_.refine_reset()
    for _ in range(max(args.warmup, 1 if stage2 else 0)):
        step()
    for nd_ in nodes:
        nd_.sync()
    for nd_ in nodes:    # HIP-event kernel timing on every context's own stream (kernels of different contexts overlap by design)
        nd_.profile(True)
        nd_.profile_reset()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    for nd_ in nodes:
        nd_.sync()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    for nd_ in nodes:
        nd_.profile(False)
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    # untimed accounting pass: published skeletons of one (steady-state) step
    s1 = [r for nd_ in nodes for r in nd_.fetch_stage1(B)]
    n_s1 = sum(r.n_persons for r in s1)
    assert not any(r.flags & pm.FLAG_HYP_OVERFLOW for r in s1), "hypothesis capacity (max_hyp) exceeded: results truncated"
    if stage2:
        per_call = []
        step(collect=per_call)
        n_skel_local = sum(per_call)
    else:
        n_skel_local = n_s1
    n_skel = n_skel_local
    if dist is not None:
        tot = torch.tensor([n_skel_local], dtype=torch.int64, device="cuda")
        dist.all_reduce(tot)
        n_skel = int(tot.item())

    if rank == 0:
        names = ["normalize", "epipolar_pairs", "sweep", "triangulate", "postprocess"] + (["track_assoc", "lm_refine", "track_update", "stage2_walk"] if stage2 else [])
        prof = {k: tuple(sum(x) for x in zip(*[nd_.profile_get(k) for nd_ in nodes])) for k in names}   # (ms, launches) over all contexts
        per_step_ms = {k: v[0] / args.steps / NC for k, v in prof.items()}                              # per context and step
        dom = max(per_step_ms, key=lambda k: per_step_ms[k])
        ms_dom = prof[dom][0] / max(prof[dom][1], 1)              # average duration of one launch (group) of the dominant kernel
        launches_per_step = prof[dom][1] / args.steps / NC
        bytes_per_skel = w["bytes_s1"] if dom in names[:5] else w["bytes_s2"]
        units_per_launch = n_skel_local / NC / launches_per_step   # skeletons one launch of that kernel processes (context 0)
        achieved = bytes_per_skel * units_per_launch / (ms_dom * 1e-3) / 1e9 if ms_dom > 0 else 0.0
        traffic = None  # HBM bytes per launch from the committed PMC passes of this exact configuration (profiles/r1/pmc_traffic_*.json)
        pmc_file = os.path.join(ROOT, "profiles", "r1", "pmc_traffic_%s.json" % args.workload)
        if os.path.exists(pmc_file) and S_ == w["seqs"] and T_ == T_STEPS and stage2:
            traffic = json.load(open(pmc_file))["kernels"].get(dom, {}).get("hbm_bytes_per_launch_corrected")
        line = {
            "metric": "3D skeletons/sec (triangulated + refined, per-joint error < 1 mm vs CPU reference path)",
            "value": round(n_skel * args.steps / dt, 2), "unit": "skeletons/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 geometry, f64 accumulation/eigen/LM", "data": "synthetic",
            "config": {"workload": "%s, stage %s, %d contexts x %d sequences x %d frames resident per GPU (%d frames, %d published skeletons per step per GPU; stage-1 persons %d)"
                                   % (args.workload, "1+2 (association+triangulation+UT covariance, then tracking+LM refinement+marginals)" if stage2 else "1",
                                      NC, S_, T_, NC * B, n_skel_local, n_s1),
                       "n_cam": w["n_cam"], "n_persons": w["n_persons"], "n_joints": 17, "frames_per_step": NC * B, "contexts_per_gpu": NC,
                       "max_hypotheses_per_frame": int(max(r.n_hyp for r in s1)), "max_published_per_frame": int(max_pub[0]),
                       "sharding": ("independent sequences per rank + RCCL all-gather of final skeletons" if world > 1 else "single GPU")},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": traffic, "avg_launch_ms": round(ms_dom, 5),
                         "algorithmic_bytes_per_skeleton": bytes_per_skel, "skeletons_per_launch": round(units_per_launch, 1),
                         "note": "SURVEY 8d: the path is VALU/latency-bound, not HBM-bound; the HBM fraction is reported as north_star asks",
                         "kernel_ms_per_step": {k: round(v, 5) for k, v in per_step_ms.items()}},
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(rig, seqs[0], node.params, stage2)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    for nd_ in nodes:
        nd_.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
