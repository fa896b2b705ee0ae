#!/usr/bin/env python3
"""bench.py — 3D skeletons/sec of the MI355X-native skeleton_3d + pose_prior hot path on synthetic scenes.

One "step" = one pass of the hot path over one batch of synthetic frames that are already resident in HBM:
  stage 1 (association + DLT triangulation + UT covariance + merge) over all B = S x T frames, then
  stage 2 (track association + LM refinement + marginals + prediction) walking the T time steps of the S independent
  sequences.  Time steps are walked forward on even steps and backward on odd steps so that person motion stays
  continuous, tracks persist and (after warm-up) every person-frame is published (num_obs > 10, PP:845).
`value` = published refined 3D skeletons (person-frames) per second over all ranks.  See DESIGN.md §5.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload 16cam6p|64cam32p|256cam128p] [--seqs S] [--stage 1|12|dlt]

The JSON line of a default run carries, next to the headline (16cam6p, the first configuration BASELINE.json's metric is quoted on):
  "secondary": the same measurement at 256cam128p (the second configuration of the metric),
  "reprojection": the feedback reprojection of the serial pass's predicted skeletons: 16 384 frame slots per launch,
  "serial":    ONE context holding all sequences, kernels strictly one after the other on one stream — the per-kernel times that
               "roofline" is computed from (the kernels of the headline's contexts overlap by design; their durations are not
               per-kernel evidence),
  "cpu_baseline": the CPU oracle timed on this box's cores (N = 1 only).
With --gpus N > 1 and no RANK in the environment bench.py launches its N ranks itself (python -m torch.distributed.run, one fresh child
process per GPU; the parent never touches the GPU); under an external torchrun it reads RANK / LOCAL_RANK / WORLD_SIZE.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import numpy as np

# Each context owns one HIP stream; the ROCm runtime multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

T_STEPS = 16  # frames per synthetic sequence (>= 16 frames per sequence)
WORKLOADS = {
    # bytes/skeleton: SURVEY §8d algorithmic bytes (stage 1); stage 2 adds 8.9 KB per skeleton.  flop_s1: SURVEY §8d minimal-work
    # flops per skeleton (association + triangulation/UT covariance, fp32-class); stage 2: 0.74 MFLOP (fp64) per LM iteration (+1 for
    # the marginals).
    "16cam6p": dict(n_cam=16, n_persons=6, seqs=512, contexts=12, max_det=8, max_hyp=12, bytes_s1=8972, bytes_s2=8900, flop_s1=0.12e6 + 1.86e6),
    # the `robust` block (round 6): the 16cam6p shape on rough scenes (scene.ROUGH: 0.6 % gross 2-D outliers, 8 % missing views, 15 % low-confidence
    # joints, 2 px noise) — the reference's outlier path (S3D:748-838) and the float32 second pass of the triangulation are hot here
    "16cam6p_rough": dict(n_cam=16, n_persons=6, seqs=512, contexts=12, max_det=8, max_hyp=16, bytes_s1=8972, bytes_s2=8900, flop_s1=0.12e6 + 1.86e6, rough=True),
    # ... and with gross outliers just above the reference's threshold ("roughly 50 pixels", S3D:59) on 4 % of the observations: ~5 % of the joints take the
    # leave-one-out rejection itself (S3D:793-838) and more sit in the threshold band — the level at which the second pass carries >= 5 % of the joints
    "16cam6p_outliers": dict(n_cam=16, n_persons=6, seqs=512, contexts=12, max_det=8, max_hyp=16, bytes_s1=8972, bytes_s2=8900, flop_s1=0.12e6 + 1.86e6, rough=dict(outlier_frac=0.04, outlier_px=60.0)),
    "64cam32p": dict(n_cam=64, n_persons=32, seqs=32, contexts=3, max_det=40, max_hyp=96, bytes_s1=30072, bytes_s2=8900, flop_s1=11.0e6 + 7.28e6),
    # diagnostic: a mid-size rig with P <= 16 and N x P > 256 — the stand-alone normalisation + the flat pair kernel (one lane per table entry) and the
    # one-wavefront sweep with its state in global memory; byte / flop figures interpolated from the two neighbours, not a bench line
    "32cam12p": dict(n_cam=32, n_persons=12, seqs=64, contexts=4, max_det=16, max_hyp=32, bytes_s1=15000, bytes_s2=8900, flop_s1=2.5e6 + 3.6e6),
    # 256cam128p: 6 contexts x 4 sequences x 16 frames; the pair table is kept as two camera-column chunks per frame (Se3dPairPlan):
    # 9.4 GB per 16 frames instead of 68.9 GB, 6 x 38.5 GB per GPU.  Contexts (profiles/r4/README.md section 7): 3 -> 26.1 k, 4 -> 26.9 - 27.3 k,
    # 5 -> 27.9 - 28.7 k skeletons/s (one more set of latency-bound sweep / Munkres / LM chains under the pair kernels), 6 with one stream per
    # context -> 28.3 - 29.2 k (with the second stream per context: 11.6 - 15.0 k), 7 -> 29.0 k (277 GB), 8 does not fit
    "256cam128p": dict(n_cam=256, n_persons=128, seqs=4, contexts=6, max_det=128, max_hyp=256, bytes_s1=115320, bytes_s2=8900, flop_s1=0.71e9 + 29.0e6),
}
FLOP_S2_PER_LM_ITER = 0.74e6
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
VALU_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: fp32 vector peak (fp64 vector: half of it)
STAGE1_KERNELS = ["normalize", "epipolar_pairs", "sweep", "triangulate", "postprocess"]
STAGE2_KERNELS = ["track_assoc", "lm_refine", "track_update", "stage2_walk"]


# ---------------------------------------------------------------------------------------------------------------------------------
# launcher: --gpus N starts N ranks (fresh children; nothing in this process has touched the GPU yet)
# ---------------------------------------------------------------------------------------------------------------------------------
def self_launch(n):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


# ---------------------------------------------------------------------------------------------------------------------------------
# synthetic scenes (generated in worker processes forked BEFORE anything touches the GPU)
# ---------------------------------------------------------------------------------------------------------------------------------
def make_rig(w):
    from smartedgesensor3dhumanpose_amd import rig as R
    from smartedgesensor3dhumanpose_amd.skeleton_3d import rig_tensors
    rig = R.extr_rig_16cam() if w["n_cam"] == 16 else R.elliptical_rig(w["n_cam"], seed=0)
    rig.set_tensors(*rig_tensors(rig))   # host code of the library (se3d_rig_from_transforms): no GPU call
    return rig


def _gen_one(job):
    n_cam, n_persons, t_steps, seed = job[:4]
    rough = job[4] if len(job) > 4 else None   # None / False: SURVEY's clean generator; True: scene.ROUGH; a dict: scene.ROUGH with these overrides
    from smartedgesensor3dhumanpose_amd import rig as R, scene as S
    rig = R.extr_rig_16cam() if n_cam == 16 else R.elliptical_rig(n_cam, seed=0)
    cov = 4 if n_cam == 16 else 6
    cfg = (S.rough_config(n_persons, t_steps, seed, min_coverage=cov, **(rough if isinstance(rough, dict) else {})) if rough
           else S.SceneConfig(n_persons=n_persons, n_frames=t_steps, seed=seed, min_coverage=cov))
    return S.generate_sequence(rig, cfg)[0]


DISTINCT_DEFAULT = 32


def make_sequences(pool, w, n_ctx, n_seq, seed_rank, t_steps, distinct=DISTINCT_DEFAULT):
    """Per context: n_seq synthetic sequences of t_steps frames — `distinct` seeded sequences per context, each resident n_seq / distinct
    times in its own memory (default 32; --distinct all: every resident sequence its own scene).  Seeds: 1000 x rank + 100 x context + i for
    the default (the scenes of every earlier round), 10^6 x (rank + 1) + n_seq x context + i when a context holds more than 100."""
    n_uniq = min(n_seq, distinct)
    base = (lambda ci, i: 1000 * seed_rank + 100 * ci + i) if n_uniq <= 100 else (lambda ci, i: 1000000 * (seed_rank + 1) + n_seq * ci + i)
    jobs = [(w["n_cam"], w["n_persons"], t_steps, base(ci, i), w.get("rough")) for ci in range(n_ctx) for i in range(n_uniq)]
    uniq = pool.map(_gen_one, jobs, chunksize=4 if len(jobs) > 1024 else 1) if pool is not None else [_gen_one(j) for j in jobs]
    return [[uniq[ci * n_uniq + s % n_uniq] for s in range(n_seq)] for ci in range(n_ctx)]


def cpu_baseline(rig, seq, params, with_stage2, budget_s=12.0):
    """Oracle (CPU restatement of the reference, kind 'port') timed on this box's host cores on a bounded sample.  Next to the rate: the
    per-stage wall-clock averages bucketed by the number of published persons, as the reference's own timers report them at node exit
    (g_timings[0] = all frames, g_timings[n] = frames with n persons, n < 10: S3D:1076-1084, 1234-1241; PP:912-920, 952-959)."""
    from oracle import oracle_py as O
    ncores = os.cpu_count() or 1
    out, buckets = {}, {}
    for label, thr in (("1t", 1), ("omp", ncores)):
        pp = O.PosePrior(params) if with_stage2 else None
        t0 = time.perf_counter()
        n_sk, n_fr = 0, 0
        bk = {}
        while time.perf_counter() - t0 < budget_s / 2:
            fr = seq[n_fr % len(seq)]
            ta = time.perf_counter()
            r = O.triangulate(rig, params, fr, capacity=512, hyp_capacity=512, omp_threads=thr)
            tb = time.perf_counter()
            n2 = 0
            if pp is not None:
                r2 = pp.step(r.published(), rig.n_cam, 100.0 + n_fr / 30.0, fr.fb_delay, capacity=512, omp_threads=thr)
                n2 = r2.n_persons
            tc = time.perf_counter()
            for key, n_, dt_ in (("stage1", r.n_persons, tb - ta), ("stage2", n2, tc - tb)):
                for b in ("all", str(n_) if n_ < 10 else ">=10"):
                    e = bk.setdefault(key, {}).setdefault(b, [0, 0.0])
                    e[0] += 1
                    e[1] += dt_
            n_sk += r.n_persons
            n_fr += 1
        dt = time.perf_counter() - t0
        out[label] = (n_sk / dt, n_fr, dt)
        buckets[label] = {k: {b: {"frames": e[0], "avg_ms": round(e[1] / e[0] * 1e3, 4)} for b, e in sorted(v.items())} for k, v in bk.items()}
    best = "omp" if out["omp"][0] > out["1t"][0] else "1t"
    return {"value": round(out[best][0], 2), "unit": "skeletons/s", "cores": ncores if best == "omp" else 1, "kind": "port",
            "sample": "%d frames of one sequence of the same workload, stage %s; single thread %.1f skel/s (%d frames), OpenMP x%d over "
                      "hypotheses/persons as in S3D:676 / PP:582: %.1f skel/s (%d frames)"
                      % (out[best][1], "1+2" if with_stage2 else "1", out["1t"][0], out["1t"][1], ncores, out["omp"][0], out["omp"][1]),
            "per_stage_ms": {"note": "wall-clock ms per frame of the oracle, bucketed by the number of persons the stage published (the reference's "
                                     "g_timings buckets, S3D:1076-1084 / PP:912-920); stage1 = triangulate_persons, stage2 = pose_prior skeletonCallback",
                             "single_thread": buckets["1t"], "openmp_x%d" % ncores: buckets["omp"]}}


def bench_parity(name, rig, seqs, n_seq=4, device=0, omp_threads=1, max_steps=None):
    """The second half of the metric (BASELINE.json: "3D skeletons/sec + mm MPJPE vs CPU ref"; SURVEY 8d: max and mean per-joint error, association
    exact-match rate), untimed: the HIP path against the CPU oracle on the bench's OWN scenes through a context of the same limits — stage 1 per
    frame (n_seq sequences x T frames), then the resident stage-2 walk against the oracle's pose_prior fed with the same stage-1 persons."""
    from oracle import oracle_py as O
    from smartedgesensor3dhumanpose_amd import person_msgs as pm
    from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
    w = WORKLOADS[name]
    seqs = [sq[:max_steps] if max_steps else sq for sq in seqs[:n_seq]]   # (omp_threads / max_steps: the 256-camera block — the oracle's sweep and triangulation over the host's cores, one frame)
    S_, T_ = len(seqs), len(seqs[0])
    frames = [seqs[s][k] for k in range(T_) for s in range(S_)]
    node = Skeleton3D(rig, max_epi_dist=0.045, max_frames=S_ * T_, max_det_per_cam=w["max_det"], max_hyp=w["max_hyp"], device=device)
    node.stage(frames)
    node.run_stage1()
    node.refine_reset()
    r1 = node.fetch_stage1(S_ * T_)
    t0 = time.perf_counter()
    exact, d_all, n_skel = 0, [], 0
    for fr, g in zip(frames, r1):
        o = O.triangulate(rig, node.params, fr, capacity=2 * w["max_hyp"], hyp_capacity=2 * w["max_hyp"], omp_threads=omp_threads)
        same = (g.flags == o.flags and g.n_hyp == o.n_hyp and g.n_persons == o.n_persons and np.array_equal(g.assoc(), o.assoc())
                and np.array_equal(g.person_hyp[:g.n_persons], o.person_hyp[:g.n_persons]))
        exact += int(same)
        if same and g.n_persons:
            gk, ok = g.published()["keypoints"], o.published()["keypoints"]
            v = ok["score"] > 0
            same_valid = np.array_equal(gk["score"] > 0, v)
            exact -= int(not same_valid)
            if same_valid:
                d_all.append(np.linalg.norm(gk["joint"] - ok["joint"], axis=-1)[v])
                n_skel += g.n_persons
    out = {"workload": name, "frames": len(frames), "skeletons": n_skel, "assoc_exact_rate": exact / len(frames)}
    if d_all:
        d = np.concatenate(d_all)
        out.update(mpjpe_max_mm=round(float(d.max() * 1e3), 6), mpjpe_mean_mm=round(float(d.mean() * 1e3), 6), joints=int(len(d)))
    # stage 2: the resident walk (the bench's default path) vs the oracle's pose_prior on the same stage-1 persons
    ts = [100.0 + (i + 1) / 30.0 for i in range(T_)]
    if w["max_hyp"] <= 16:   # the kernel form the headline runs (one wavefront per sequence, three persons per LM wavefront); few sequences would default to the workgroup form
        node.set_options(walk_shape="wave")
    node.run_stage2_walk(np.repeat(np.array(ts)[:, None], S_, 1), [k * S_ for k in range(T_)])
    node.sync()
    oras = [O.PosePrior(node.params) for _ in range(S_)]
    d2, ids_ok, it_ok, steps2, n_ref = [], 0, 0, 0, 0
    for k, t in enumerate(ts):
        ra = node.fetch_stage2_walk(k * S_, S_)
        for s_ in range(S_):
            o2 = oras[s_].step(r1[k * S_ + s_].published(), rig.n_cam, t, seqs[s_][k].fb_delay, capacity=2 * w["max_hyp"], omp_threads=omp_threads)
            g2 = ra[s_]
            steps2 += 1
            n = g2.n_persons
            ok_ids = n == o2.n_persons and np.array_equal(g2.fused["id"][:n], o2.fused["id"][:n])
            ids_ok += int(ok_ids)
            if ok_ids and n:
                it_ok += int(np.array_equal(g2.lm_iterations[:n], o2.lm_iterations[:n]))
                for nm in ("fused", "fused_pred"):
                    gk, ok = getattr(g2, nm)[:n]["keypoints"], getattr(o2, nm)[:n]["keypoints"]
                    v = ok["score"] > 0
                    if np.array_equal(gk["score"] > 0, v):
                        d2.append(np.linalg.norm(gk["joint"] - ok["joint"], axis=-1)[v])
                n_ref += n
            elif ok_ids:
                it_ok += 1
    out.update(stage2_steps=steps2, refined_skeletons=n_ref, track_ids_exact_rate=ids_ok / max(steps2, 1), lm_iterations_equal_rate=it_ok / max(steps2, 1))
    if d2:
        d = np.concatenate(d2)
        out.update(refined_max_mm=round(float(d.max() * 1e3), 6), refined_mean_mm=round(float(d.mean() * 1e3), 6))
    out["oracle_seconds"] = round(time.perf_counter() - t0, 2)
    node.close()
    return out


def bench_latency_b1(rig, seq, name, device=0, frames_timed=200):
    """The reference's operating point: ONE synchronised frame per callback (S3D:1069, PP:505).  se3d_triangulate + se3d_refine of one 16cam6p
    frame through the synchronous host calls (pageable host buffers in, results out) and the device time of the kernels in between
    (HIP events on the context's stream); frames of a sequence in order, so tracks persist and every person is refined."""
    from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
    w = WORKLOADS[name]
    node = Skeleton3D(rig, max_epi_dist=0.045, max_frames=1, max_det_per_cam=w["max_det"], max_hyp=w["max_hyp"], device=device)
    node.refine_reset()
    T_ = len(seq)
    order = list(range(T_)) + list(range(T_ - 2, 0, -1))    # forward and back: continuous motion

    def run(n, t_start):
        t1 = t2 = 0.0
        n_ref = 0
        for i in range(n):
            fr = seq[order[i % len(order)]]
            a = time.perf_counter()
            r = node.triangulate_persons([fr])[0]
            b = time.perf_counter()
            f = node.refine(0, r.published(), t_start + (i + 1) / 30.0, fr.fb_delay)
            c = time.perf_counter()
            t1 += b - a
            t2 += c - b
            n_ref += f.n_persons
        return t1 / n, t2 / n, n_ref / n
    run(40, 100.0)
    # host-call times as a caller sees them: kernel timing OFF, so each call is one hipGraph replay (upload + kernel chain + download, round 4)
    h1, h2, n_ref = run(frames_timed, 100.0 + 41 / 30.0)
    # device times per kernel group: kernel timing ON (HIP events between the kernels: the calls then launch directly, no graph)
    node.profile(True)
    node.profile_reset()
    p1, p2, _ = run(frames_timed, 100.0 + (41 + frames_timed) / 30.0)
    node.profile(False)
    dev = {k: node.profile_get(k) for k in STAGE1_KERNELS + STAGE2_KERNELS}
    d1 = sum(dev[k][0] for k in STAGE1_KERNELS) / frames_timed
    d2 = sum(dev[k][0] for k in STAGE2_KERNELS) / frames_timed
    node.close()
    return {"workload": "%s, one frame per call (B = 1)" % name, "frames_timed": frames_timed, "refined_skeletons_per_frame": round(n_ref, 2),
            "host_call_ms": {"se3d_triangulate": round(h1 * 1e3, 4), "se3d_refine": round(h2 * 1e3, 4), "frame": round((h1 + h2) * 1e3, 4)},
            "host_call_ms_direct_launches": {"se3d_triangulate": round(p1 * 1e3, 4), "se3d_refine": round(p2 * 1e3, 4), "frame": round((p1 + p2) * 1e3, 4)},
            "device_ms": {"stage1": round(d1, 4), "stage2": round(d2, 4), "frame": round(d1 + d2, 4),
                          "by_kernel_group": {k: round(dev[k][0] / frames_timed, 4) for k in STAGE1_KERNELS + STAGE2_KERNELS}},
            "note": "host_call_ms: synchronous C-ABI calls incl. H2D / D2H of pageable buffers, each call one hipGraph replay (options.graph = -1: direct launches); "
                    "device_ms: HIP events around the kernel groups of directly launched calls (the second pass, host_call_ms_direct_launches; reference call sites "
                    "S3D:1069, PP:505); the oracle's ms per frame: cpu_baseline.per_stage_ms"}


# ---------------------------------------------------------------------------------------------------------------------------------
# one measurement: NC contexts x S sequences x T frames resident on this rank's GPU
# ---------------------------------------------------------------------------------------------------------------------------------
def dense_slots_per_frame(w):
    """Person capacity per frame of the dense exchange blocks (se3d_pack_stage2_walk_device_dense)."""
    return w["n_persons"] + 0.25 if w["n_persons"] <= 16 else 1.125 * w["n_persons"] + 2


def memory_plan(name, n_ctx, n_frames, exchange=False, free_bytes=None):
    """Device memory one rank's GPU needs for `n_ctx` contexts of `n_frames` resident frames each (se3d_device_bytes: the library's own
    allocation list, host arithmetic) + the exchange's pack / gather buffers; raises SystemExit with the figures when it exceeds what is
    free, instead of a hipMalloc error half-way through the context creation."""
    from smartedgesensor3dhumanpose_amd import person_msgs as pm
    from smartedgesensor3dhumanpose_amd.skeleton_3d import device_bytes
    w = WORKLOADS[name]
    per_ctx = device_bytes(w["n_cam"], n_frames, w["max_det"], w["max_hyp"])
    ex = 0
    if exchange:   # SkeletonExchange: 2 pack + 2 gather buffers of [n_ctx][n_frames][max_hyp] persons (+ counts), gather x world
        world = int(os.environ.get("WORLD_SIZE", "1"))
        slots = dense_slots_per_frame(w)   # dense exchange blocks
        ex = int(2 * (1 + world) * n_ctx * n_frames * (slots * pm.person3d_dt.itemsize + 4))
    plan = {"workload": name, "contexts": n_ctx, "frames_per_context": n_frames, "bytes_per_context": per_ctx, "exchange_bytes": ex,
            "total_bytes": n_ctx * per_ctx + ex}
    if free_bytes is not None and plan["total_bytes"] > free_bytes:
        raise SystemExit("bench.py: %s with %d contexts x %d frames needs %.1f GB of device memory per GPU (%.2f GB per context%s), %.1f GB are free: "
                         "lower --contexts / --seqs" % (name, n_ctx, n_frames, plan["total_bytes"] / 1e9, per_ctx / 1e9,
                                                        ", %.2f GB exchange buffers" % (ex / 1e9) if ex else "", free_bytes / 1e9))
    return plan


def run_workload(name, seqs_per_ctx, rig, args, steps, warmup, dist, local_rank, stage2, verify_exchange=True, reproj=False):
    """seqs_per_ctx: list over contexts of lists of sequences.  Returns a dict (all ranks; `value` etc. are whole-job figures)."""
    import torch
    from smartedgesensor3dhumanpose_amd import person_msgs as pm
    from smartedgesensor3dhumanpose_amd import shard
    from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
    w = WORKLOADS[name]
    NC, S_, T_ = len(seqs_per_ctx), len(seqs_per_ctx[0]), len(seqs_per_ctx[0][0])
    B = S_ * T_
    world = dist.get_world_size() if dist is not None else 1
    memory_plan(name, NC, B, exchange=dist is not None, free_bytes=torch.cuda.mem_get_info(local_rank)[0])   # fails fast, with the figures
    nodes = []
    # More than four contexts of a large rig: ONE stream per context (se3d_options.hi_priority_stream = -1 at se3d_create_ex).  The second,
    # high-priority stream a large-rig context owns for its sweep chains buys nothing once this many contexts interleave (5 contexts: 28.6 k
    # without, 27.9 - 28.7 k with), and 2 x 6 streams collapse to 11.6 - 15.0 k skeletons/s (profiles/r4/README.md section 7).
    one_stream = w["n_cam"] >= 128 and NC > 4 and not args.hi_streams
    for seqs in seqs_per_ctx:   # every rank (and every context) owns its own sequences: weak scaling, no data-path collective
        frames = [seqs[s][k] for k in range(T_) for s in range(S_)]   # frame slot = k * S + s
        nd_ = Skeleton3D(rig, max_epi_dist=0.045, max_frames=B, max_det_per_cam=w["max_det"], max_hyp=w["max_hyp"], device=local_rank,
                         options=dict(**(dict(hi_priority_stream=-1) if one_stream else {}), **(dict(pairs_form=args.pairs_form) if getattr(args, "pairs_form", None) else {})) or None)
        nd_.stage(frames)  # inputs resident in HBM before the timed region
        nodes.append(nd_)
    node = nodes[0]
    dt_frame = 1.0 / 30.0
    H = node.limits.max_hyp
    psz = pm.person3d_dt.itemsize
    fused_walk = args.walk == "fused"
    ex = None
    if dist is not None:
        # Exchange: ONE RCCL all-gather of the final skeletons per step so that every rank holds the skeletons of all ranks
        # ([contexts][B][XH] person slots + [contexts][B] counts, frame-slot order); stream-ordered and double-buffered
        # (shard.SkeletonExchange, covered by tests/test_distributed_gloo.py with two gloo ranks).
        # person slots per frame carried by the exchange: crowded scenes publish spurious extra persons (256cam128p: up to ~150 per frame),
        # so only the small workloads trim the slots below the hypothesis capacity (checked: truncation is an assertion, not a silent loss)
        XH = min(H, w["n_persons"] + 2) if stage2 and fused_walk and w["n_persons"] <= 16 else H   # slot form (--exchange-slots, large workloads)
        # Small workloads with the fused walk: the DENSE form (se3d_pack_stage2_walk_device_dense) — the published persons back to back +
        # per-frame offsets, capacity n_persons + 0.25 persons per frame; what does not fit is counted in the block, checked below on every
        # rank for every rank's block.  At 16cam6p a rank produces ~51 GB/s of final skeletons; the all-gather delivers (world - 1) x that
        # into every GPU, so padding slots ([frame][n_persons + 2] before: +28 %) decide whether the exchange hides behind the step at 8 GPUs.
        # Round 6: the dense form also at 256cam128p ([frame][256 slots] carried 2 x padding over the <= 132 persons a frame publishes): capacity
        # 1.125 x n_persons + 2 per frame there (crowded scenes publish spurious extra persons; what does not fit is counted and asserted to be zero)
        dense_cap = int(math.ceil(B * dense_slots_per_frame(w))) if stage2 and fused_walk and not args.exchange_slots else 0
        streams = [torch.cuda.ExternalStream(nd_.stream_handle()) for nd_ in nodes]
        ex = shard.SkeletonExchange(dist, NC, (dense_cap if dense_cap else B * XH) * psz, B + 2 if dense_cap else B, torch.device("cuda", local_rank), streams)

    # One step = per context one stage-1 pass over its B resident frames and one stage-2 walk over the T time steps of its S
    # sequences (se3d_run_stage2_walk: one launch, a wavefront or workgroup per sequence; forward on even cycles, backward on odd
    # ones so that motion stays continuous and tracks persist).  With --walk steps the walk is issued as T per-step calls.
    # Contexts are phase-shifted (odd contexts issue walk-then-stage-1) so that the throughput-bound stage-1 kernels of one context
    # overlap the latency-bound stage-2 walk of another instead of all contexts running the same phase in lockstep.  Stage-1
    # outputs of a frame slot are a pure function of the resident inputs, so a walk always consumes complete stage-1 results.
    walks = [{"node": nd_, "ci": ci, "cycle": 0, "t": 100.0} for ci, nd_ in enumerate(nodes)]
    max_pub = [0]   # most skeletons any frame published (accounting pass): the exchange carries n_persons + 2 slots per frame
    lm_iters = []

    # Large association problems (256 cameras): the pair-cost kernel is VALU-bound and fills the GPU, the camera sweep and the stage-2
    # steps behind it are chains of small latency-bound launches.  With all contexts in the same phase the GPU alternates between
    # saturated and nearly idle; chaining the pair kernels of the contexts by events (one at a time, round robin) staggers the contexts so
    # that the latency-bound chains of two contexts run under the pair kernel of the third.
    chain = w["n_cam"] >= 128 and NC > 1 and not args.no_chain
    if chain:
        tstreams = [torch.cuda.ExternalStream(nd_.stream_handle()) for nd_ in nodes] if ex is None else streams
        pair_done = [None] * NC

    def issue_stage1(wk):
        if chain:
            ci = wk["ci"]
            prev = pair_done[(ci - 1) % NC]
            if prev is not None:
                tstreams[ci].wait_event(prev)
            wk["node"].run_stage1_phases(1)
            pair_done[ci] = tstreams[ci].record_event()
            wk["node"].run_stage1_phases(8 | 2 | 4)
        else:
            wk["node"].run_stage1()
        if not stage2 and ex is not None:
            pk, cn = ex.pack_target(wk["ci"])
            wk["node"].pack_stage1_device(pk.data_ptr(), cn.data_ptr())
            ex.packed(wk["ci"])

    def issue_walk(wk, collect):
        nd_, ci = wk["node"], wk["ci"]
        ks = list(range(T_)) if wk["cycle"] % 2 == 0 else list(range(T_ - 1, -1, -1))
        ts = wk["t"] + dt_frame * np.arange(1, T_ + 1)
        wk["t"] = float(ts[-1])
        wk["cycle"] += 1
        if fused_walk:
            nd_.run_stage2_walk(np.repeat(ts[:, None], S_, 1), [k * S_ for k in ks])
            if ex is not None:  # async D2D of all time steps' fused skeletons (frame-slot order) into the exchange buffer
                pk, cn = ex.pack_target(ci)
                if dense_cap:
                    nd_.pack_stage2_walk_device_dense(B, dense_cap, pk.data_ptr(), cn.data_ptr())
                else:
                    nd_.pack_stage2_walk_device_slots(B, XH, pk.data_ptr(), cn.data_ptr())
                ex.packed(ci)
            if collect is not None:
                res = [r for k in ks for r in nd_.fetch_stage2_walk(k * S_, S_)]
                per_frame = [r.n_persons for r in res]
                assert ex is None or dense_cap or max(per_frame) <= XH, "exchange slots per frame too small: skeletons would be truncated"
                max_pub[0] = max(max_pub[0], max(per_frame))
                collect.append(sum(per_frame))
                if ci == 0:
                    lm_iters.extend(int(v) for r in res[: 4 * S_] for v in r.lm_iterations[: r.n_persons])
        else:
            pk, cn = ex.pack_target(ci) if ex is not None else (None, None)
            for k, t in zip(ks, ts):
                nd_.run_stage2(np.full(S_, t), frame_slot0=k * S_)
                if ex is not None:
                    nd_.pack_stage2_device(S_, pk.data_ptr() + k * S_ * H * psz, cn.data_ptr() + 4 * k * S_)
                if collect is not None:
                    res = nd_.fetch_stage2(S_)
                    collect.append(sum(r.n_persons for r in res))
                    max_pub[0] = max([max_pub[0]] + [r.n_persons for r in res])
            if ex is not None:
                ex.packed(ci)

    def step(collect=None, exchange=True):
        if not stage2:
            for wk in walks:
                issue_stage1(wk)
        else:
            for wk in walks:   # phase A
                issue_stage1(wk) if wk["ci"] % 2 == 0 else issue_walk(wk, collect)
            for wk in walks:   # phase B
                issue_walk(wk, collect) if wk["ci"] % 2 == 0 else issue_stage1(wk)
        return ex.exchange() if ex is not None and exchange else None

    for nd_ in nodes:
        nd_.run_stage1()   # phase-shifted contexts start inside the stage-2 walk: their frames need one completed stage-1 pass
    for nd_ in nodes:
        nd_.refine_reset()
    for _ in range(max(warmup, 1 if stage2 else 0)):
        step()
    for nd_ in nodes:
        nd_.sync()
    for nd_ in nodes:    # HIP-event kernel timing on every context's own stream
        nd_.profile(True)
        nd_.profile_reset()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        last_parity = step()
    for nd_ in nodes:
        nd_.sync()
    torch.cuda.synchronize()   # includes the exchange stream and RCCL's stream
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    for nd_ in nodes:
        nd_.profile(False)
    if dist is not None:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    exchange_ms = dt_nx = None
    if ex is not None:
        ex.finish()
    exchange_ok = None
    if ex is not None and verify_exchange:   # every rank's gathered buffer == concatenation of all ranks' packs (last timed step)
        ok = torch.tensor([1 if ex.verify(last_parity) else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        exchange_ok = bool(ok.item())
        assert exchange_ok, "RCCL exchange: a gathered buffer differs from the ranks' pack buffers"
    # ---- diagnosis of the exchange (round 4; the scaling curve is the driver's to measure, these keys say where a loss comes from): the
    # all-gather's own duration per step (stamped above) and the SAME step with the all-gather skipped (packs still written), one short pass ----
    if ex is not None:
        steps_nx = max(2, steps // 4)
        dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(steps_nx):
            step(exchange=False)
        for nd_ in nodes:
            nd_.sync()
        torch.cuda.synchronize()
        dist.barrier()
        dt_nx = (time.perf_counter() - t1) / steps_nx
        tnx = torch.tensor([dt_nx], dtype=torch.float64, device="cuda")
        dist.all_reduce(tnx, op=dist.ReduceOp.MAX)
        dt_nx = float(tnx.item())
        # ... and the all-gather's own duration, stamped with events on the exchange stream in a pass of its own (the stamps add
        # event records and device-side waits, which do not belong into the headline's timed region)
        ex.timed(True)
        dist.barrier()
        for _ in range(steps_nx):
            step()
        for nd_ in nodes:
            nd_.sync()
        torch.cuda.synchronize()
        ex.finish()
        exchange_ms = ex.exchange_ms()
        ex.timed(False)
        dist.barrier()
    # untimed accounting pass: published skeletons of one (steady-state) step
    s1 = [r for nd_ in nodes for r in nd_.fetch_stage1(B)]
    n_s1 = sum(r.n_persons for r in s1)
    n_overflow = sum(1 for r in s1 if r.flags & pm.FLAG_HYP_OVERFLOW)
    if not args.ablation:   # ablation: kernel-timing builds that write wrong tables on purpose (tools/tmp)
        # (rough scenes stress the capacity by nature: a frame whose sweep spawns more hypotheses than max_hyp is truncated by contract and COUNTED — the
        # clean workloads must never overflow)
        assert w.get("rough") or n_overflow == 0, "hypothesis capacity (max_hyp) exceeded: results truncated"
        assert not any(r.flags & pm.FLAG_ASSIGNMENT_FAILED for r in s1), "the sweep's Hungarian hit its iteration guard"
    if stage2:
        per_call = []
        step(collect=per_call)
        n_skel_local = sum(per_call)
        if ex is not None:
            ex.finish()
            _, gc = ex.gathered(ex.parity ^ 1)   # gathered counts of the accounting step: nothing truncated on any rank
            if dense_cap:   # [world][contexts][B + 2]: word B = persons written, word B + 1 = persons that did not fit
                assert int(gc[:, :, B + 1].max().item()) == 0, "a rank's dense exchange block was too small: skeletons were dropped"
                assert int(gc[dist.get_rank(), :, B].sum().item()) == n_skel_local, "the dense exchange blocks do not hold this rank's published skeletons"
            else:
                assert int(gc.max().item()) <= XH, "a remote rank published more skeletons in a frame than the exchange carries"
    else:
        n_skel_local = n_s1
    n_skel = n_skel_local
    if dist is not None:
        tot = torch.tensor([n_skel_local], dtype=torch.int64, device="cuda")
        dist.all_reduce(tot)
        n_skel = int(tot.item())
    # joints the triangulation's float32 second pass redid in the last stage-1 launch of context 0 (round 4: se3d_get_exact_list)
    _xl, _x64, _ = nodes[0].exact_list(flags=True)
    n_second_pass, n_second_cov64 = len(_xl), int(_x64.sum())
    names = STAGE1_KERNELS + ["triangulate_exact"] + (STAGE2_KERNELS if stage2 else [])
    prof = {k: tuple(sum(x) for x in zip(*[nd_.profile_get(k) for nd_ in nodes])) for k in names}   # (ms, launches) over all contexts
    res = {
        "workload": name, "value": n_skel * steps / dt, "ms_per_step": dt / steps * 1e3, "n_skel": n_skel, "n_skel_local": n_skel_local,
        "n_s1": n_s1, "NC": NC, "S": S_, "T": T_, "B": B, "steps": steps, "warmup": warmup, "prof": prof,
        "kernel_ms_per_step": {k: v[0] / steps / NC for k, v in prof.items()},   # per context and step
        "max_hyp_per_frame": int(max(r.n_hyp for r in s1)), "frames_hyp_overflow": n_overflow, "max_published_per_frame": int(max_pub[0]), "exchange_verified": exchange_ok,
        "lm_iterations_mean": float(np.mean(lm_iters)) if lm_iters else None, "params": node.params, "pair_chunks": node.pair_chunks(),
        "second_pass_joints": n_second_pass, "second_pass_fraction": n_second_pass / max(1.0, 17.0 * n_s1 / NC), "second_pass_cov64": n_second_cov64,
        "sweep_stats": nodes[0].sweep_stats(),   # (camera steps of the per-camera sweep with hypotheses and detections, those that ran the Hungarian): large rigs
        "exchange": None if ex is None else {"form": "dense: published persons back to back + per-frame offsets" if dense_cap else "[frame][%d person slots] + counts" % XH,
                                             "bytes_per_rank_per_step": int(NC * (ex.sp + 4 * ex.sc))},
        "exchange_ms_per_step": None if exchange_ms is None else round(exchange_ms[0], 4), "exchange_ms_max": None if exchange_ms is None else round(exchange_ms[1], 4),
        "ms_per_step_no_exchange": None if dt_nx is None else dt_nx * 1e3, "value_no_exchange": None if dt_nx is None else n_skel / dt_nx,
    }
    if reproj and stage2 and fused_walk:
        res["reprojection"] = bench_reprojection(node, rig, min(B, 16384))
    for nd_ in nodes:
        nd_.close()
    return res


def bench_reprojection(node, rig, n_slots, launches=5):
    """SURVEY 8f rank 1, measured: the feedback reprojection (pose_reprojection, REPROJ:139-235) of the predicted skeletons that the walk
    left in HBM — n_slots frame slots per launch (se3d_reproject_resident_batch), every skeleton into every camera (7 sigma points per
    joint, fp64), per-camera Person2D lists compacted on the device.  Throughput from `launches` back-to-back launches (host clock around
    a stream synchronisation: the three kernels of a launch run for milliseconds); output volume from a sample of 64 slots; CPU: the
    oracle's reprojection (kind 'port', single thread) of the same sample."""
    from oracle import oracle_py as O
    node.set_reprojection_cameras()
    node.reproject_resident_batch(0, n_slots)   # allocates the batch buffers, warm-up
    node.sync()
    t0 = time.perf_counter()
    for _ in range(launches):
        node.reproject_resident_batch(0, n_slots)
    node.sync()
    dt = (time.perf_counter() - t0) / launches
    sample = list(range(0, n_slots, max(1, n_slots // 64)))[:64]
    n_sk = n_out = 0
    preds = []
    for slot in sample:
        r = node.fetch_stage2_walk(slot, 1)[0]
        n_sk += r.n_persons
        n_out += int(node.fetch_reprojection(slot).det_offset[-1])
        preds.append(r.fused_pred[:r.n_persons].copy())
    scale = n_slots / len(sample)
    n_sk_launch, n_out_launch = n_sk * scale, n_out * scale
    t1 = time.perf_counter()
    for p in preds:
        O.reproject(rig, "simple", p)
    cpu_dt = time.perf_counter() - t1
    from smartedgesensor3dhumanpose_amd import person_msgs as pm
    bytes_launch = n_sk_launch * pm.person3d_dt.itemsize + n_out_launch * pm.person2d_dt.itemsize
    return {"value": round(n_sk_launch / dt, 1), "unit": "skeletons/s reprojected into all %d cameras" % rig.n_cam, "ms_per_launch": round(dt * 1e3, 4),
            "slots_per_launch": n_slots, "skeletons_per_launch": round(n_sk_launch), "projected_persons_per_launch": round(n_out_launch),
            "roofline": {"bound": "hbm", "achieved": round(bytes_launch / dt / 1e9, 2), "peak": 8000.0, "unit": "GB/s", "frac": round(bytes_launch / dt / 8e12, 5),
                         "algorithmic_bytes": "1688 B per skeleton in + 428 B per (skeleton, camera that sees it) out",
                         "flop_note": "fp64 VALU: ~0.5 kFLOP per (joint, camera); %.1f GFLOP/s" % (n_sk_launch * rig.n_cam * 17 * 500 / dt / 1e9)},
            "cpu_baseline": {"value": round(n_sk / cpu_dt, 1) if cpu_dt > 0 else None, "unit": "skeletons/s", "cores": 1, "kind": "port",
                             "sample": "%d skeletons of %d slots" % (n_sk, len(sample))}}


def bench_dlt(args):
    """Config 2 of BASELINE.json / SURVEY 8d: 16 cam x 6 p x 17 joints, batched DLT triangulation only (rows N, T, P, O, U), input = a
    known association (here: the table the library's own sweep produced for the same frames, computed once, untimed).
    B = 4096 frames per launch for throughput (`value`), B = 1 for latency (device-only and through the synchronous host call)."""
    import torch   # noqa: F401  (device runtime of the image; the bench needs a GPU)
    from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
    name = args.workload or "16cam6p"
    w = WORKLOADS[name]
    rig = make_rig(w)
    B = args.seqs or 4096
    seqs = make_sequences(None, w, 1, min(B, 32), 0, args.tsteps)[0]
    frames = [seqs[i % len(seqs)][(i // len(seqs)) % args.tsteps] for i in range(B)]
    node = Skeleton3D(rig, max_epi_dist=0.045, max_frames=B, max_det_per_cam=w["max_det"], max_hyp=w["max_hyp"], device=0)
    node.stage(frames)
    node.run_stage1()
    s1 = node.fetch_stage1(B)
    H = w["max_hyp"]
    assoc = np.full((B, H, rig.n_cam), -1, np.int32)
    for k, r in enumerate(s1):
        assoc[k, : r.n_hyp] = r.assoc()
    node.stage_assoc(assoc)
    for _ in range(max(args.warmup, 1)):
        node.run_dlt()
    node.sync()
    node.profile(True)
    node.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        node.run_dlt()
    node.sync()
    dt = time.perf_counter() - t0
    node.profile(False)
    ms, n = node.profile_get("dlt")
    out = node.fetch_dlt()
    n_skel = int(((out["keypoints"]["score"] > 0).sum(-1) > 0).sum())
    node.close()
    # latency: one frame
    one = Skeleton3D(rig, max_epi_dist=0.045, max_frames=1, max_det_per_cam=w["max_det"], max_hyp=H, device=0)
    for _ in range(20):
        one.triangulate_assoc(frames[:1], assoc[:1])
    t1 = time.perf_counter()
    for _ in range(200):
        one.triangulate_assoc(frames[:1], assoc[:1])
    lat_host = (time.perf_counter() - t1) / 200
    one.stage(frames[:1]); one.stage_assoc(assoc[:1])
    one.profile(True); one.profile_reset()
    for _ in range(200):
        one.run_dlt()
    one.sync()
    lat_ms, lat_n = one.profile_get("dlt")
    one.close()
    avg_ms = ms / max(n, 1)
    achieved = w["bytes_s1"] * n_skel / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    line = {"metric": "3D skeletons/sec, batched DLT triangulation + UT covariance only (BASELINE config 2: association given)",
            "value": round(n_skel * args.steps / dt, 2), "unit": "skeletons/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 rows, f64 Gram / null vector", "data": "synthetic",
            "config": {"workload": "%s, config 2 (rows N,T,P,O,U), B = %d frames per launch, %d hypotheses triangulated per launch" % (name, B, n_skel),
                       "n_cam": w["n_cam"], "n_persons": w["n_persons"], "n_joints": 17},
            "latency_B1": {"device_ms": round(lat_ms / max(lat_n, 1), 5), "host_call_ms": round(lat_host * 1e3, 4),
                           "note": "device: HIP events around the three launches of one frame; host: synchronous se3d_triangulate_assoc incl. H2D / D2H"},
            "roofline": {"bound": "hbm", "kernel": "dlt (normalize + association table + triangulate)", "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None, "avg_launch_ms": round(avg_ms, 5),
                         "algorithmic_bytes_per_skeleton": w["bytes_s1"], "skeletons_per_launch": n_skel,
                         "valu": {"algorithmic_flop_per_skeleton": 1.86e6, "achieved": round(1.86e6 * n_skel / (avg_ms * 1e-3) / 1e12, 4) if avg_ms > 0 else 0.0,
                                  "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s"}},
            "cpu_baseline": None}
    print(json.dumps(line), flush=True)


def gpu_local_cpus(device):
    """The CPUs of the NUMA node the GPU's PCIe root hangs on (sysfs `local_cpulist` of its PCI function), restricted to the CPUs this process may
    use; None when that cannot be read."""
    try:
        import torch
        p = torch.cuda.get_device_properties(device)
        txt = open("/sys/bus/pci/devices/%04x:%02x:%02x.0/local_cpulist" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)).read().strip()
        cpus = set()
        for part in txt.split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        return cpus or None
    except Exception:
        return None


def numa_nodes_of(addr, nbytes, samples=8):
    """NUMA node of `samples` pages spread over [addr, addr + nbytes): move_pages(2) in query mode (x86-64 syscall 279); None when the kernel refuses."""
    import ctypes
    try:
        libc = ctypes.CDLL(None, use_errno=True)
        out = []
        for i in range(samples):
            page = ctypes.c_void_p((addr + (nbytes * i) // samples) & ~4095)
            status = ctypes.c_int(-1000)
            if libc.syscall(279, 0, 1, ctypes.byref(page), None, ctypes.byref(status), 0) != 0:
                return None
            out.append(int(status.value))
        return out
    except Exception:
        return None


def run_streaming(name, seqs_per_ctx, rig, steps, warmup, local_rank, stream_out=0, affinity="local", local_source=True):
    """PCIe-inclusive rate (never `value`): every step each context receives its batch of frames from HOST memory (se3d_stream_submit:
    ragged payload into pinned memory, H2D on a copy stream, expansion on the device, stage 1, stage-2 walk, device compaction of the
    published persons, D2H) and hands the published fused / predicted persons back in host memory (se3d_stream_collect); two batches per
    context in flight, one host thread per context (the C calls release the GIL)."""
    import threading
    from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
    w = WORKLOADS[name]
    NC, S_, T_ = len(seqs_per_ctx), len(seqs_per_ctx[0]), len(seqs_per_ctx[0][0])
    B = S_ * T_
    nodes, prepared = [], []
    in_bytes = 0
    diag = {}
    # host side of a host-fed deployment: the feeding threads (and the pinned buffers they first touch) on the NUMA node of the GPU's PCIe root, so
    # that neither the copy into pinned memory nor the DMA out of it crosses the socket interconnect (affinity="none": wherever the scheduler puts them)
    all_cpus, local = os.sched_getaffinity(0), (gpu_local_cpus(local_rank) if affinity == "local" else None)
    if local:
        os.sched_setaffinity(0, local)
    try:   # (an exception in the streaming pass must not leave the later passes pinned to one NUMA node)
        from smartedgesensor3dhumanpose_amd.person_msgs import Frame
        for seqs in seqs_per_ctx:
            frames = [seqs[s][k] for k in range(T_) for s in range(S_)]
            # Where the SOURCE frames live (round 6).  The scene generator's results were unpickled by the main thread wherever the scheduler had it at the
            # time; the feeding threads run on the GPU's NUMA node.  With the frames on the other node, about a third of the runs of rounds 4 - 6 showed ONE
            # step of 83 - 95 ms among ~32 ms ones, on every context at once, ~0.7 s into the pass, and faster steps after it (profiles/r6/README.md
            # section 3: the signature of the kernel's automatic NUMA balancing moving the pages to the node that reads them).  A host-fed deployment
            # receives its messages on the node it runs on; the pass copies the frames once from a local CPU (this thread's affinity is already set).
            if local_source and local:   # ONE fresh block per context (a new mapping: its pages are first touched here), the frames are views into it
                blob = np.empty(sum(len(f.persons) for f in frames), frames[0].persons.dtype)
                o, local_frames = 0, []
                for f in frames:
                    blob[o:o + len(f.persons)] = f.persons
                    local_frames.append(Frame(f.n_cam, np.array(f.det_offset, copy=True), blob[o:o + len(f.persons)], np.array(f.fb_delay, copy=True), f.stamp))
                    o += len(f.persons)
                frames = local_frames
            if "source_numa" not in diag and len(frames) and len(frames[0].persons):
                diag["source_numa"] = numa_nodes_of(frames[0].persons.ctypes.data, frames[0].persons.nbytes, 2)
            nd_ = Skeleton3D(rig, max_epi_dist=0.045, max_frames=B, max_det_per_cam=w["max_det"], max_hyp=w["max_hyp"], device=local_rank, options=dict(stream_out=stream_out))
            nd_.refine_reset()
            nodes.append(nd_)
            prepared.append(nd_.stream_prepare(frames))
            in_bytes += sum(int(f.det_offset[-1]) for f in frames) * 428
        counts = [[0, 0] for _ in nodes]   # published persons, bytes out per context (timed region)
        step_ms = []
        slowest = [[-1, 0.0, 0.0] for _ in nodes]
        diag["slowest"] = slowest
        go, done = threading.Barrier(NC + 1), threading.Barrier(NC + 1)

        def worker(ci):
            nd_, t, cyc = nodes[ci], 100.0, 0

            def submit():
                nonlocal t, cyc
                ks = list(range(T_)) if cyc % 2 == 0 else list(range(T_ - 1, -1, -1))
                ts = t + (1.0 / 30.0) * np.arange(1, T_ + 1)
                t, cyc = float(ts[-1]), cyc + 1
                nd_.stream_submit(prepared[ci], np.repeat(ts[:, None], S_, 1), [k * S_ for k in ks])
            submit()
            for _ in range(warmup):
                submit()
                nd_.stream_collect(copy=False)
            go.wait()
            t_prev = time.perf_counter()
            for _ in range(steps):
                ta = time.perf_counter()
                submit()
                tb = time.perf_counter()
                off, fused, _pred = nd_.stream_collect(copy=False)
                t_end = time.perf_counter()
                if (t_end - ta) * 1e3 > slowest[ci][1]:
                    slowest[ci] = [_, round((t_end - ta) * 1e3, 1), round((tb - ta) * 1e3, 1)]   # step index, whole step ms, of it submit ms
                if ci == 0:   # host time of the two calls (context 0): submit = copy into pinned memory + enqueue, collect = wait for the oldest batch + its download
                    diag.setdefault("submit_ms", []).append((tb - ta) * 1e3)
                    diag.setdefault("collect_ms", []).append((time.perf_counter() - tb) * 1e3)
                counts[ci][0] += int(off[-1])
                counts[ci][1] += 2 * int(off[-1]) * 1688 + 4 * len(off)
                if ci == 0:   # context 0's step times: tells a uniformly slow run (both copy directions at ~60 %: one run in ten, cause open) from a transient
                    if "numa" not in diag and len(fused):   # where the context's pinned output buffer lives, and where this thread runs
                        diag["numa"] = numa_nodes_of(fused.ctypes.data, fused.nbytes)
                        diag["cpu"] = os.sched_getcpu() if hasattr(os, "sched_getcpu") else None
                    t_now = time.perf_counter()
                    step_ms.append(round((t_now - t_prev) * 1e3, 2))
                    t_prev = t_now
            nd_.stream_collect(copy=False)
            done.wait()
        threads = [threading.Thread(target=worker, args=(ci,)) for ci in range(NC)]
        for th in threads:
            th.start()
        go.wait()
        t0 = time.perf_counter()
        done.wait()
        dt = time.perf_counter() - t0
        try:
            diag["numa_balancing"] = int(open("/proc/sys/kernel/numa_balancing").read())
        except (OSError, ValueError):
            diag["numa_balancing"] = None
        for th in threads:
            th.join()
        diag["probe"] = nodes[0].copy_stream_probe()   # untimed, after the measurement: do the device's two copy streams (still) overlap?
        for nd_ in nodes:
            nd_.close()
    finally:
        if local:
            os.sched_setaffinity(0, all_cpus)
    n_skel = sum(c[0] for c in counts)
    return {"value": n_skel / dt, "ms_per_step": dt / steps * 1e3, "contexts": NC, "sequences_per_context": S_, "steps": steps,
            "host_threads": ("on the %d CPUs of the GPU's NUMA node" % len(local)) if local else "unpinned", "step_ms_context0": step_ms,
            "pinned_output_numa_nodes_context0": diag.get("numa"), "source_frames_numa_nodes": diag.get("source_numa"), "worker_cpu_context0": diag.get("cpu"),
            "kernel_numa_balancing": diag.get("numa_balancing"), "copy_stream_probe": diag.get("probe"),
            "host_call_ms_context0": {"submit_mean": round(float(np.mean(diag.get("submit_ms", [0]))), 2), "collect_mean": round(float(np.mean(diag.get("collect_ms", [0]))), 2),
                                      "submit_max": round(float(np.max(diag.get("submit_ms", [0]))), 2), "collect_max": round(float(np.max(diag.get("collect_ms", [0]))), 2),
                                      "slowest_step": int(np.argmax(step_ms)) if step_ms else None,
                                      "slowest_step_per_context": diag.get("slowest")},
            "h2d_gb_per_s": in_bytes * steps / dt / 1e9, "d2h_gb_per_s": sum(c[1] for c in counts) / dt / 1e9,
            "h2d_mb_per_step": in_bytes / 1e6, "skeletons_per_step": n_skel // steps}


def guarded(fn, seconds, what):
    """Run `fn` on a worker thread with a deadline: (result, error text or None, True if it is still running)."""
    import threading
    box = {}

    def run():
        try:
            box["r"] = fn()
        except BaseException as e:   # noqa: BLE001 — reported in the JSON line
            box["e"] = "%s: %s" % (type(e).__name__, e)
    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(seconds)
    if th.is_alive():
        return None, "%s did not finish within %.0f s" % (what, seconds), True
    return box.get("r"), box.get("e"), False


def run_hypothesis_shard(name, seq, rig, steps, dist, local_rank):
    """north_star config 5 as a latency mode: ONE stream of frames, every rank holds the same frames, association and track lifecycle
    replicated, triangulation and LM refinement split by index % world, RCCL all-gathers of the hypothesis / detection blocks
    (shard.HypothesisShard).  Verified live: the sharded results of a whole walk equal the un-sharded results of the same context bit
    for bit on every rank.  Only rows T/P/O/U and F/R/Z shard, so the speed-up over one GPU is bounded by their share of the step."""
    import torch
    from smartedgesensor3dhumanpose_amd import shard
    from smartedgesensor3dhumanpose_amd.skeleton_3d import Skeleton3D
    w = WORKLOADS[name]
    T_ = len(seq)
    nd_ = Skeleton3D(rig, max_epi_dist=0.045, max_frames=T_, max_det_per_cam=w["max_det"], max_hyp=w["max_hyp"], device=local_rank)
    nd_.stage(seq)   # frame slot k = time step k of the one sequence
    ts = 100.0 + (1.0 / 30.0) * np.arange(1, T_ + 1)

    def walk(hs):
        out = []
        if hs is None:
            nd_.run_stage1()
        else:
            hs.run_stage1()
        for k in range(T_):
            if hs is None:
                nd_.run_stage2(ts[k:k + 1], frame_slot0=k)
            else:
                hs.run_stage2(ts[k:k + 1], k)
            r = nd_.fetch_stage2(1)[0]
            out.append((r.n_persons, r.c.n_tracks, r.fused[: r.n_persons].tobytes(), r.fused_pred[: r.n_persons].tobytes()))
        return out, [r.published().tobytes() for r in nd_.fetch_stage1(T_)]
    nd_.refine_reset()
    ref = walk(None)
    hs = shard.HypothesisShard(dist, nd_, T_, 1, torch.device("cuda", local_rank))
    nd_.refine_reset()
    got = walk(hs)
    ok = torch.tensor([1 if got == ref else 0], dtype=torch.int32, device="cuda")
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    n_pub = sum(g[0] for g in got[0])
    # timed: steady-state walks (tracks persist; forward only, stamps advance)
    t_off = [float(ts[-1])]

    def step():
        hs.run_stage1()
        tt = t_off[0] + (1.0 / 30.0) * np.arange(1, T_ + 1)
        t_off[0] = float(tt[-1])
        for k in range(T_):
            hs.run_stage2(tt[k:k + 1], k)
    step()
    nd_.sync()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    nd_.sync()
    torch.cuda.synchronize()
    dist.barrier()
    dt = time.perf_counter() - t0
    tmax = torch.tensor([dt], dtype=torch.float64, device="cuda")
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    n_last = sum(r.n_persons for r in [nd_.fetch_stage2(1)[0]]) * T_   # persons of the last step x steps (steady state: constant per frame)
    nd_.set_shard(0, 1)
    nd_.close()
    return {"verified_equal_to_single_rank": bool(ok.item()), "ms_per_frame": float(tmax.item()) / steps / T_ * 1e3, "frames_per_step": T_,
            "skeletons_per_s": n_last * steps / float(tmax.item()), "published_in_verification_walk": n_pub,
            "allgather_bytes_per_step": hs.bytes_per_step}


def roofline_of(res, w, stage2, source):
    """HBM roofline of the dominant kernel of a measurement: achieved = algorithmic bytes per skeleton x skeletons per launch / average
    launch duration (HIP events on the kernel's own stream)."""
    prof, NC, steps = res["prof"], res["NC"], res["steps"]
    per_step_ms = res["kernel_ms_per_step"]
    dom = max(per_step_ms, key=lambda k: per_step_ms[k])
    ms_dom = prof[dom][0] / max(prof[dom][1], 1)              # average duration of one launch (group) of the dominant kernel
    launches_per_step = prof[dom][1] / steps / NC
    bytes_per_skel = w["bytes_s1"] if dom in STAGE1_KERNELS else w["bytes_s2"]
    units_per_launch = res["n_skel_local"] / NC / max(launches_per_step, 1e-9)
    achieved = bytes_per_skel * units_per_launch / (ms_dom * 1e-3) / 1e9 if ms_dom > 0 else 0.0
    # `bound`: the resource the committed PMC passes show as binding for this kernel (profiles/r5, r6 READMEs) — none of the large kernels of this
    # path is HBM-bound (~9 KB of algorithmic bytes per skeleton against ~2 MFLOP); achieved / peak / frac remain the HBM figures the
    # contract asks for (`frac_basis`)
    binding = {"triangulate": "valu (fp64 vector issue; HBM at %.0f %% of peak)", "epipolar_pairs": "valu+lds (fp32 vector issue and the LDS read pipe; HBM at %.0f %% of peak)",
               "stage2_walk": "valu latency (dependent fp64 chain; HBM at %.0f %% of peak)"}.get(dom, "hbm (%.0f %% of peak)") % (100.0 * achieved / HBM_PEAK_GBS)
    return {"bound": binding, "frac_basis": "hbm: algorithmic bytes per launch / launch time / 8 TB/s", "kernel": dom, "achieved": round(achieved, 3), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 6), "traffic": None, "avg_launch_ms": round(ms_dom, 5),
            "algorithmic_bytes_per_skeleton": bytes_per_skel, "skeletons_per_launch": round(units_per_launch, 1), "source": source,
            "sum_kernel_ms_per_step": round(sum(per_step_ms.values()), 4),
            "kernel_ms_per_step": {k: round(v, 5) for k, v in per_step_ms.items()}}


def valu_of(res, w, stage2, pmc, mix=None, kernel=None, live_ms=None):
    """Vector-ALU roofline (SURVEY 8d asks for both): algorithmic flops per skeleton x skeletons/s against the fp32 vector peak."""
    it = res.get("lm_iterations_mean") or 0.0
    flop = w["flop_s1"]
    tf = flop * res["value_per_gpu"] / 1e12
    out = {"algorithmic_flop_per_skeleton": round(flop), "achieved": round(tf, 4), "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / VALU_PEAK_TFLOPS, 6),
           "lm_iterations_mean": round(it, 3),
           "stage2_dense_equivalent_flop_per_skeleton": round(FLOP_S2_PER_LM_ITER * (it + 1.0)) if stage2 else 0,
           "note": "SURVEY 8d minimal-work flops of stage 1 (association + triangulation + UT covariance) x skeletons/s against the fp32 vector peak; the "
                   "kernels issue more (IEEE-exact divides, fp64 null-vector solves).  Stage 2 is not in `achieved`: SURVEY counts it as dense 63x63 normal "
                   "equations (0.74 MFLOP per LM iteration), which the forest elimination never forms (DESIGN 3.1)"}
    if pmc:   # committed SQ counter pass (tools/profile_r2.sh, tools/pmc_sq.py): VALU wave-instructions x 2 issue cycles / (1024 SIMDs x kernel cycles)
        ks = [k for n_, k in pmc["kernels"].items() if "SQ_INSTS_VALU" in k["counters_per_launch"] and "reproj" not in n_]   # the step: stage 1 + stage 2
        cyc = sum(k["kernel_cycles"] * k["launches"] for k in ks)
        if cyc > 0:
            out["measured_valu_issue_utilisation"] = round(sum(k["counters_per_launch"]["SQ_INSTS_VALU"] * k["launches"] for k in ks) * 2.0 / (1024.0 * cyc), 4)
            out["measured_source"] = pmc.get("_file", "profiles") + ": " + pmc.get("note", "")
    if mix and mix[0] and mix[1]:   # committed instruction-mix passes (tools/profile_r3.sh valumix, tools/valumix.py): the dominant kernel priced by type
        a, b = mix[0]["kernels"].get(kernel), mix[1]["kernels"].get(kernel)
        if a and b:
            c = dict(a["counters_per_launch"])
            c.update(b["counters_per_launch"])
            g = lambda n_: c.get("SQ_INSTS_VALU_" + n_, 0.0)
            m = {"f64": g("ADD_F64") + g("MUL_F64") + g("FMA_F64"), "trans_f64": g("TRANS_F64"), "f32": g("ADD_F32") + g("MUL_F32") + g("FMA_F32"), "trans_f32": g("TRANS_F32"),
                 "cvt": g("CVT"), "int32": g("INT32"), "int64": g("INT64")}
            m["other"] = c.get("SQ_INSTS_VALU", 0.0) - sum(m.values())
            cost = {"f64": 4, "trans_f64": 16, "f32": 2, "trans_f32": 8, "cvt": 4, "int32": 2, "int64": 4, "other": 2}   # issue cycles of a wave64 instruction on a SIMD
            cyc = a["kernel_cycles"]
            out["dominant_kernel_issue_model"] = {
                "kernel": kernel, "issue_frac": round(sum(m[x] * cost[x] for x in m) / 1024.0 / cyc, 4),
                "fp64_share_of_kernel_cycles": round(m["f64"] * 4 / 1024.0 / cyc, 4),
                "fp64_tflops": round((g("ADD_F64") + g("MUL_F64") + 2 * g("FMA_F64")) * 64 / (a["avg_us"] * 1e-6) / 1e12, 2), "fp64_vector_peak_tflops": 78.6,
                "wave_instructions_M": {x: round(m[x] / 1e6) for x in m},
                "source": mix[0].get("_file", "profiles") + " + " + mix[1].get("_file", "profiles") + " (vector-ALU instruction counters of the serial pass, priced at their issue cost)"}
            # The figure to quote: what the dominant kernel MEASURABLY does on the vector ALU — its fp64 operations per launch (PMC
            # instruction counters of the committed pass; the counts do not change from run to run) over its LIVE launch time, against the fp64 vector
            # peak.  SURVEY's minimal-work flop model moves to `survey_minimal_work_model`: for T + U it exceeds the fp64 peak at these rates, i.e.
            # the kernels do far fewer flops than that model assumes, so a fraction of it says nothing about the kernel.
            if live_ms:   # the dominant kernel's launch time of THIS run's serial pass
                fl = (g("ADD_F64") + g("MUL_F64") + 2 * g("FMA_F64")) * 64
                out["survey_minimal_work_model"] = {k_: out.pop(k_) for k_ in ("algorithmic_flop_per_skeleton", "achieved", "peak", "unit", "frac", "note")}
                out.update({"bound": "fp64 vector ALU (dominant kernel: %s)" % kernel, "achieved": round(fl / (live_ms * 1e-3) / 1e12, 3), "peak": 78.6, "unit": "TFLOP/s fp64",
                            "frac": round(fl / (live_ms * 1e-3) / 1e12 / 78.6, 4), "issue_utilisation": out["dominant_kernel_issue_model"]["issue_frac"],
                            "note": "measured: fp64 add / mul / fma wave-instructions of the dominant kernel per launch (committed PMC pass) x 64 lanes / its live launch "
                                    "time, against the 78.6 TFLOP/s fp64 vector peak; issue_utilisation prices every VALU instruction of the kernel at its issue cost"})
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS), help="default: 16cam6p headline + 256cam128p secondary + serial pass")
    ap.add_argument("--seqs", type=int, default=0, help="independent sequences per context (default: workload preset)")
    ap.add_argument("--tsteps", type=int, default=T_STEPS)
    ap.add_argument("--stage", default="12", choices=["1", "12", "dlt"], help="1: stage 1 only; 12: stage 1 + stage 2; dlt: config 2 "
                    "(batched DLT triangulation + UT covariance with the association given)")
    ap.add_argument("--contexts", type=int, default=0, help="independent se3d contexts (HIP streams) per GPU, each with --seqs sequences; "
                    "their launches interleave so latency-bound phases of one overlap throughput-bound kernels of the other (default: preset)")
    ap.add_argument("--walk", default="fused", choices=["fused", "steps"], help="stage 2 as one multi-step launch per context and step (default) "
                    "or as T per-step calls (A/B)")
    ap.add_argument("--serial", action="store_true", help="ONLY the serial pass: one context with contexts x seqs sequences, one stream "
                    "(the run rocprofv3 --kernel-trace --stats is pointed at for profiles/*/serial_*)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the 256cam128p secondary measurement of a default run")
    ap.add_argument("--no-serial", action="store_true", help="skip the serial per-kernel pass of a default run")
    ap.add_argument("--no-streaming", action="store_true", help="skip the PCIe-inclusive streaming measurement (value_streaming) of a default run")
    ap.add_argument("--stream-out", type=int, default=0, choices=[0, 1], help="se3d_options.stream_out of the streaming measurement: 0 = device buffers + D2H copies (default), "
                    "1 = the compaction kernel writes the published persons into pinned host memory")
    ap.add_argument("--stream-affinity", choices=["local", "none"], default="local", help="streaming pass: host threads on the GPU's NUMA node (sysfs local_cpulist) or unpinned")
    ap.add_argument("--stream-source", choices=["local", "as-is"], default="local", help="streaming pass: the frames the feeding threads read are copied once, by a thread on the GPU's "
                    "NUMA node, before the pass (local: default) or left where the scene generator's results were unpickled (as-is: diagnostic)")
    ap.add_argument("--stream-prefree-gb", type=float, default=0.0, help="diagnostic: allocate, touch and free this much device memory right before the streaming pass")
    ap.add_argument("--stream-settle-s", type=float, default=-1.0, help="seconds to wait between the last freed context and the streaming pass (default: 3 s in a default run, "
                    "0 otherwise; profiles/r6/README.md section 3)")
    ap.add_argument("--stream-repeat", type=int, default=1, help="diagnostic: repeat the streaming measurement this many times in the process (each with fresh contexts); "
                    "every result goes to stderr, the last into the line")
    ap.add_argument("--stream-contexts", type=int, default=8, help="contexts of the streaming measurement (each pins two batches of host memory)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rough", default="", help="diagnostic: overrides of scene.ROUGH for the rough-scene workload, e.g. pixel_noise=3.5,outlier_frac=0.012")
    ap.add_argument("--pairs-form", default=None, choices=["flat", "rows", "fused"], help="diagnostic: se3d_options.pairs_form of every context (default: the library's choice)")
    ap.add_argument("--no-robust", action="store_true", help="skip the `robust` block of a default run (the 16cam6p shape on rough scenes: scene.ROUGH)")
    ap.add_argument("--distinct", default=str(DISTINCT_DEFAULT), help="distinct seeded sequences per context (default %d, each resident seqs / %d times); "
                    "'all': every resident sequence its own scene (control run; generation takes ~10 s on 32 processes)" % (DISTINCT_DEFAULT, DISTINCT_DEFAULT))
    ap.add_argument("--latency-only", action="store_true", help="only the B = 1 latency block (one 16cam6p frame per call), printed as its own JSON line")
    ap.add_argument("--no-parity", action="store_true", help="skip the untimed parity / latency blocks of a default run")
    ap.add_argument("--exchange-backend", default="nccl", choices=["nccl", "gloo-staged"], help="nccl (= RCCL over xGMI, default) | gloo-staged: TEST ONLY — device "
                    "buffers staged through host memory and exchanged by gloo, so that the N > 1 code path can run with several ranks on one GPU")
    ap.add_argument("--hi-streams", action="store_true", help="keep the second (high-priority) stream of large-rig contexts also with more than four contexts per GPU")
    ap.add_argument("--exchange-slots", action="store_true", help="diagnostic: exchange the small workloads in the [frame][slot] form instead of the dense form")
    ap.add_argument("--no-chain", action="store_true", help="diagnostic: do not chain the pair kernels of the contexts by events at 256 cameras")
    ap.add_argument("--ablation", action="store_true", help="diagnostic: kernel-timing builds that write wrong tables on purpose (tools/tmp): skip the result checks")
    ap.add_argument("--seed-rank", type=int, default=None, help="diagnostic: run another rank's synthetic scenes on this GPU")
    ap.add_argument("--secondary-last", action="store_true", help="measure 256cam128p after the headline instead of first")
    ap.add_argument("--debug-mem", action="store_true")
    ap.add_argument("--hyp-deadline-s", type=float, default=900.0, help="deadline of the hypothesis-shard block (a collective that never returns ends the run with exit code 3)")
    ap.add_argument("--no-parity-256", action="store_true", help="skip the one-frame 256cam128p oracle comparison of the parity block")
    ap.add_argument("--hyp-shard", action="store_true", help="with --workload 256cam128p under the launcher: also run the hypothesis-shard block "
                    "(north_star config 5 as a latency mode) that a default multi-rank run ends with")
    args = ap.parse_args()

    if args.rough:   # before the scene workers are forked: they read scene.ROUGH
        from smartedgesensor3dhumanpose_amd import scene as _sc
        _sc.ROUGH.update({k: float(v) for k, v in (kv.split("=") for kv in args.rough.split(","))})
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(self_launch(args.gpus))   # the children print the JSON line (rank 0)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if "RANK" in os.environ and world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.stage == "dlt":
        return bench_dlt(args)

    if args.latency_only:   # A/B of the synchronous single-frame calls: same scene and block as the default run's latency_B1
        w = WORKLOADS["16cam6p"]
        print(json.dumps({"latency_B1": bench_latency_b1(make_rig(w), _gen_one((w["n_cam"], w["n_persons"], args.tsteps, 0)), "16cam6p", local_rank)}), flush=True)
        return 0
    default_run = args.workload is None and not args.serial
    headline = args.workload or "16cam6p"
    stage2 = args.stage == "12"
    w = WORKLOADS[headline]
    S_, T_ = args.seqs or w["seqs"], args.tsteps
    NC = args.contexts or w["contexts"]
    seed_rank = rank if args.seed_rank is None else args.seed_rank  # diagnostic: run another rank's synthetic scenes on this GPU (tools/check_rank_scenes.sh)

    # ---- scenes first: worker processes are forked before this process initialises the GPU ----
    import multiprocessing as mp
    nproc = max(1, min(32, (os.cpu_count() or 1) // max(world, 1)))
    pool = mp.get_context("fork").Pool(nproc) if nproc > 1 else None
    t_gen = time.perf_counter()
    rig = make_rig(w)
    distinct = S_ if args.distinct == "all" else max(1, int(args.distinct))
    seqs_head = make_sequences(pool, w, NC, S_, seed_rank, T_, distinct)
    want_robust = default_run and stage2 and rank == 0 and not args.no_robust
    if want_robust:
        seqs_rough = make_sequences(pool, WORKLOADS["16cam6p_rough"], NC, S_, seed_rank, T_, distinct)
        seqs_outl = make_sequences(pool, WORKLOADS["16cam6p_outliers"], NC, S_, seed_rank, T_, distinct) if not args.no_serial else None
    want_secondary = default_run and not args.no_secondary and stage2
    if want_secondary:
        w2 = WORKLOADS["256cam128p"]
        rig2 = make_rig(w2)
        seqs_sec = make_sequences(pool, w2, w2["contexts"], w2["seqs"], seed_rank, T_)
    want_parity = default_run and stage2 and rank == 0 and not args.no_parity
    if want_parity:   # the parity block's 64cam32p scene (two frames of one sequence)
        rig64 = make_rig(WORKLOADS["64cam32p"])
        seq64 = (pool.apply(_gen_one, ((64, 32, 2, 77),)) if pool is not None else _gen_one((64, 32, 2, 77)))
    if pool is not None:
        pool.close()
        pool.join()
    t_gen = time.perf_counter() - t_gen

    import torch
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run (also with one rank: exercises the exchange path)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.exchange_backend == "gloo-staged":   # TEST ONLY: several ranks on one GPU (RCCL refuses that): shard.HostStagedDist
            from smartedgesensor3dhumanpose_amd.shard import HostStagedDist
            local_rank = local_rank % torch.cuda.device_count()
            torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo")
            assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
            dist = HostStagedDist(dist)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the product path has no CPU fallback")

    sec, hyp, sec_serial = None, None, None

    def run_secondary():
        nonlocal sec, hyp, sec_serial
        # steps: the four contexts take turns on the pair-cost kernel, so a timed region ends with a drain of ~100 ms (the last context's
        # sweep / stage-2 chain) that K steps of ~260 ms have to carry: K = 10 by default
        sec = run_workload("256cam128p", seqs_sec, rig2, args, max(5, args.steps // 2), 2, dist, local_rank, True)
        sec["value_per_gpu"] = sec["value"] / world
        if rank == 0 and not args.no_serial:
            # serial pass of the secondary configuration: ONE context, one stream, nothing else on the GPU -> per-kernel durations
            sec_serial = run_workload("256cam128p", [seqs_sec[0]], rig2, args, 3, 1, None, local_rank, True)
            sec_serial["value_per_gpu"] = sec_serial["value"]

    # Order of a default run: the 256cam128p measurement FIRST, on the fresh device (its six contexts take 231 of the 288 GB — 272 in round 2, whole pair tables; behind the
    # headline's twelve contexts the same measurement came out ~8 % lower, behind the streaming contexts at half speed — measured, cause
    # not isolated), then the headline, the serial pass and the streaming pass.  --secondary-last restores the old order.
    sec_first = want_secondary and not args.secondary_last
    if sec_first:
        run_secondary()
    head = None
    if not args.serial:
        head = run_workload(headline, seqs_head, rig, args, args.steps, args.warmup, dist, local_rank, stage2)
        head["value_per_gpu"] = head["value"] / world
    # ---- serial pass: ONE context with all NC x S sequences on one stream; no other context runs -> per-kernel durations that do not
    # overlap anything (rank 0's GPU only; the other ranks wait at the next barrier) ----
    serial = None
    if (args.serial or (default_run and not args.no_serial)) and rank == 0:
        all_seqs = [s for ctx_seqs in seqs_head for s in ctx_seqs]
        serial = run_workload(headline, [all_seqs], rig, args, args.steps if args.serial else max(3, args.steps // 4), args.warmup if args.serial else 1,
                              None, local_rank, stage2, reproj=True)
        serial["value_per_gpu"] = serial["value"]
    # ---- `robust` (round 6): the same 12-context shape on rough scenes (scene.ROUGH) — gross 2-D outliers, missing views, low-confidence joints:
    # the reference's outlier path (S3D:748-838) and the triangulation's float32 second pass are hot; then one serial pass of the same scenes for the
    # per-kernel times (rank 0's GPU only, like the serial pass above) ----
    robust = robust_serial = robust_outl = None
    if want_robust:
        robust = run_workload("16cam6p_rough", seqs_rough, rig, args, max(5, args.steps // 2), 2, None, local_rank, True)
        if not args.no_serial:
            robust_serial = run_workload("16cam6p_rough", [[s for ctx_seqs in seqs_rough for s in ctx_seqs]], rig, args, 3, 1, None, local_rank, True)
            robust_outl = run_workload("16cam6p_outliers", [[s for ctx_seqs in seqs_outl for s in ctx_seqs]], rig, args, 3, 1, None, local_rank, True)
            del seqs_outl
        del seqs_rough
    seqs_stream = seqs_head[: max(1, min(args.stream_contexts, NC))]
    seqs_par = seqs_head[0][:4]
    del seqs_head
    if args.debug_mem:
        free, total = torch.cuda.mem_get_info()
        print("free HBM after the headline / serial passes: %.1f of %.1f GB" % (free / 1e9, total / 1e9), file=sys.stderr)
    if want_secondary and not sec_first:
        run_secondary()
    streaming = None
    if default_run and not args.no_streaming and stage2 and rank == 0:
        if args.stream_prefree_gb > 0:   # diagnostic: what a big free right before the pass does to it
            blocks = [torch.empty(int(1e9), dtype=torch.uint8, device="cuda").fill_(1) for _ in range(int(args.stream_prefree_gb))]
            torch.cuda.synchronize()
            del blocks
            torch.cuda.empty_cache()
        # The passes before this one have just freed ~100 GB of device memory (a default run: the contexts of the headline, the serial and the robust passes),
        # which the driver wipes in the background.  Measured (profiles/r6/README.md section 3): a streaming pass that starts right behind a 150 GB free shows
        # ~100 ms stalls every third step in one run of six (6.2 M instead of 11.1 M) and single 83 - 95 ms steps in a third of the default runs of rounds
        # 4 - 6; behind a 3 s pause: none in six.  The pass measures the streaming path, not the previous pass's clean-up: it waits 3 s in a default run.
        settle = args.stream_settle_s if args.stream_settle_s >= 0 else (3.0 if default_run else 0.0)
        if settle > 0:
            torch.cuda.synchronize()
            time.sleep(settle)
        for rep in range(max(1, args.stream_repeat)):
            streaming = run_streaming(headline, seqs_stream, rig, max(3, args.steps), 4, local_rank, args.stream_out, args.stream_affinity, local_source=args.stream_source == "local")   # ~0.7 s timed at the default 20 steps
            if args.stream_repeat > 1:
                print("streaming pass %d: %.2f M skeletons/s, H2D %.1f GB/s, D2H %.1f GB/s, step ms %s, pinned output on NUMA nodes %s, copy-stream probe %s"
                      % (rep, streaming["value"] / 1e6, streaming["h2d_gb_per_s"], streaming["d2h_gb_per_s"], streaming["step_ms_context0"],
                         streaming["pinned_output_numa_nodes_context0"], dict(streaming["copy_stream_probe"], **streaming["host_call_ms_context0"])), file=sys.stderr, flush=True)
    # ---- latency mode of north_star config 5, LAST and guarded: one 256cam128p stream with its hypotheses sharded over the ranks.  Every
    # measurement of the line above is already taken; an exception here is reported in the block, and a collective that does not return
    # within the deadline (ranks != 0 wait in it while rank 0 runs its serial / streaming passes: minutes at most) ends the run with the
    # line printed instead of hanging the launcher.
    hyp_error, hyp_hung = None, False
    want_hyp = (want_secondary or (args.hyp_shard and headline == "256cam128p")) and dist is not None
    if want_hyp:
        rig_h = rig2 if want_secondary else rig
        def hyp_run():
            torch.cuda.set_device(local_rank)   # the current device is per thread
            return run_hypothesis_shard("256cam128p", _gen_one((256, 128, T_, 0)), rig_h, 2, dist, local_rank)
        hyp, hyp_error, hyp_hung = guarded(hyp_run, args.hyp_deadline_s, "hypothesis shard")
    if rank == 0:
        main_res = head or serial
        def load(fn):   # committed counter summaries of THIS build: the kernels' instruction counts are properties of the code, so a summary is used only
            # when it carries the fingerprint of the kernel sources it was collected on (tools/pmc_sq.py, _lib.source_hash) — else the
            # line says which file was skipped and falls back to the model figures
            from smartedgesensor3dhumanpose_amd._lib import source_hash
            for rd in ("r6", "r5", "r4", "r3", "r2"):
                p = os.path.join(ROOT, "profiles", rd, fn)
                if os.path.exists(p):
                    d = json.load(open(p))
                    d["_file"] = "profiles/%s/%s" % (rd, fn)
                    if source_hash() is None or d.get("source_hash") != source_hash():   # (None: a package without its sources — nothing matches)
                        stale_pmc.append(d["_file"])
                        return None
                    return d
            return None
        stale_pmc = []
        roof_src = serial or head
        roof = roofline_of(roof_src, w, stage2, "serial pass: one context, one stream, %d sequences x %d frames (%.0f skeletons/s alone)"
                           % (roof_src["S"], roof_src["T"], roof_src["value"]) if serial else "overlapped contexts (no serial pass in this run)")
        tr = load("pmc_traffic_%s_serial.json" % headline)   # committed PMC passes of `bench.py --serial` (tools/profile_bench.sh)
        if tr and serial and serial["S"] == tr.get("sequences"):
            roof["traffic"] = tr["kernels"].get(roof["kernel"], {}).get("hbm_bytes_per_launch_corrected")
        roof["valu"] = valu_of(main_res, w, stage2, load("pmc_sq_%s_serial.json" % headline),
                               (load("pmc_valumix64_%s_serial.json" % headline), load("pmc_valumix32_%s_serial.json" % headline)), roof.get("kernel"), roof.get("avg_launch_ms") if serial else None)
        line = {
            "metric": "3D skeletons/sec (triangulated + refined) + mm MPJPE vs the CPU oracle (`parity`), 16cam/6p (`value`) and 256cam/128p (`secondary`)",
            "value": round(main_res["value"], 2), "unit": "skeletons/s", "n_gpus": world, "steps": main_res["steps"], "warmup": main_res["warmup"],
            "ms_per_step": round(main_res["ms_per_step"], 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 geometry, f64 accumulation/eigen/LM", "data": "synthetic",
            "config": {"workload": "%s, stage %s, %d contexts x %d sequences x %d frames resident per GPU (%d frames, %d published skeletons per step per GPU; stage-1 persons %d); "
                                   "%d distinct seeded sequences per context, each resident %d times in its own memory"
                                   % (headline, "1+2 (association+triangulation+UT covariance, then tracking+LM refinement+marginals)" if stage2 else "1",
                                      main_res["NC"], main_res["S"], main_res["T"], main_res["NC"] * main_res["B"], main_res["n_skel_local"], main_res["n_s1"],
                                      min(main_res["S"], distinct), max(1, main_res["S"] // min(main_res["S"], distinct))),
                       "distinct_sequences_per_context": min(main_res["S"], distinct),
                       "n_cam": w["n_cam"], "n_persons": w["n_persons"], "n_joints": 17, "frames_per_step": main_res["NC"] * main_res["B"],
                       "contexts_per_gpu": main_res["NC"], "max_hypotheses_per_frame": main_res["max_hyp_per_frame"],
                       "max_published_per_frame": main_res["max_published_per_frame"], "scene_generation_s": round(t_gen, 1),
                       "sharding": ("independent sequences per rank + %s all-gather of final skeletons (verified: %s)" % ("RCCL" if args.exchange_backend == "nccl" else "gloo-staged (test-only)", main_res["exchange_verified"])
                                    if dist is not None else "single GPU"),
                       "exchange": main_res.get("exchange")},
            "roofline": roof,
            "pmc_summaries_skipped": stale_pmc,   # committed counter summaries collected on OTHER kernel sources than this build's (not used)
            "second_pass": {"joints_per_launch": main_res["second_pass_joints"], "per_published_joint": round(main_res["second_pass_fraction"], 6),
                            "of_them_with_fp64_sample_covariance": main_res["second_pass_cov64"],
                            "note": "joints the fast fp64 triangulation handed to se3d_k_triangulate_exact (the oracle's restatement of the reference's float32 JacobiSVD arithmetic, "
                                    "bit-identical to that oracle): ill-conditioned systems, reprojection errors near / above the outlier threshold, wide covariances; "
                                    "its time is inside the `triangulate` kernel group"},
        }
        if main_res["sweep_stats"][0]:   # large rigs (per-camera sweep launches): how often the sweep ran the Hungarian
            line["sweep_steps_running_the_hungarian"] = {"steps": main_res["sweep_stats"][0], "hungarian": main_res["sweep_stats"][1],
                                                         "fraction": round(main_res["sweep_stats"][1] / main_res["sweep_stats"][0], 6)}
        if dist is not None:   # what lets an N > 1 run be read (round 4): the all-gather's own time, the step without it, what RCCL reports
            line["exchange_ms_per_step"] = main_res["exchange_ms_per_step"]
            line["exchange_ms_max"] = main_res["exchange_ms_max"]
            line["value_no_exchange"] = None if main_res["value_no_exchange"] is None else round(main_res["value_no_exchange"], 2)
            line["ms_per_step_no_exchange"] = None if main_res["ms_per_step_no_exchange"] is None else round(main_res["ms_per_step_no_exchange"], 4)
            line["collective"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                                  "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None,
                                  "note": "exchange_ms_per_step = mean device time of the step's all-gather pair on its own stream (it overlaps the next step's "
                                          "kernels), stamped in a pass of its own of %d steps; value_no_exchange = the same steps with the all-gather skipped, %d extra steps" % (max(2, main_res["steps"] // 4), max(2, main_res["steps"] // 4))}
        if head is not None:
            line["overlapped_kernel_ms_per_step"] = {k: round(v, 5) for k, v in head["kernel_ms_per_step"].items()}
        if serial is not None and head is not None:
            line["serial"] = {"value": round(serial["value"], 2), "ms_per_step": round(serial["ms_per_step"], 4), "steps": serial["steps"],
                              "sequences": serial["S"], "n_gpus": 1}
        if serial is not None and serial.get("reprojection"):
            line["reprojection"] = serial["reprojection"]   # the stage after the path, on the serial pass's resident results
        if robust is not None:
            from smartedgesensor3dhumanpose_amd import scene as _scene
            rs = robust_serial
            line["robust"] = {"workload": "16cam6p shape on ROUGH scenes (%s; scene.rough_config), stage 1+2, %d contexts x %d sequences x %d frames resident per GPU, max_hyp %d"
                                          % (", ".join("%s %g" % kv for kv in sorted(_scene.ROUGH.items())), robust["NC"], robust["S"], robust["T"], WORKLOADS["16cam6p_rough"]["max_hyp"]),
                              "value": round(robust["value"], 2), "unit": "skeletons/s", "n_gpus": 1, "ms_per_step": round(robust["ms_per_step"], 4), "steps": robust["steps"],
                              "fraction_of_clean_value": round(robust["value"] / main_res["value_per_gpu"], 4), "published_skeletons_per_step": robust["n_skel_local"],
                              "stage1_persons_per_step": robust["n_s1"], "second_pass_fraction": round(robust["second_pass_fraction"], 6),
                              "second_pass_joints_per_launch_context0": robust["second_pass_joints"],
                              "overlapped_kernel_ms_per_step": {k: round(v, 5) for k, v in robust["kernel_ms_per_step"].items()},
                              "serial": None if rs is None else {"value": round(rs["value"], 2), "ms_per_step": round(rs["ms_per_step"], 4), "sequences": rs["S"],
                                                                 "second_pass_fraction": round(rs["second_pass_fraction"], 6), "second_pass_joints_per_launch": rs["second_pass_joints"], "of_them_with_fp64_sample_covariance": rs["second_pass_cov64"],
                                                                 "second_pass_ms_per_step": round(rs["kernel_ms_per_step"]["triangulate_exact"], 5),
                                                                 "kernel_ms_per_step": {k: round(v, 5) for k, v in rs["kernel_ms_per_step"].items()}},
                              "outlier_heavy_serial": None if robust_outl is None else {
                                  "workload": "the same with gross outliers of 60 px (the reference's threshold: roughly 50 px, S3D:59) on 4 %% of the observations; one context, %d sequences" % robust_outl["S"],
                                  "value": round(robust_outl["value"], 2), "ms_per_step": round(robust_outl["ms_per_step"], 4),
                                  "fraction_of_the_clean_serial_value": None if serial is None else round(robust_outl["value"] / serial["value"], 4),
                                  "second_pass_fraction": round(robust_outl["second_pass_fraction"], 6), "second_pass_joints_per_launch": robust_outl["second_pass_joints"],
                                  "of_them_with_fp64_sample_covariance": robust_outl["second_pass_cov64"], "frames_hyp_overflow": robust_outl["frames_hyp_overflow"],
                                  "second_pass_ms_per_step": round(robust_outl["kernel_ms_per_step"]["triangulate_exact"], 5),
                                  "kernel_ms_per_step": {k: round(v, 5) for k, v in robust_outl["kernel_ms_per_step"].items()}},
                              "frames_hyp_overflow": robust["frames_hyp_overflow"],
                              "note": "the headline's scenes are SURVEY's clean generator (second_pass.per_published_joint of the joints take the float32 second pass); here "
                                      "second_pass_fraction of them do, as in the randomised parity sweep's scenes; `triangulate_exact` (se3d_k_triangulate_exact alone) is also inside `triangulate`"}
        if streaming is not None:
            line["value_streaming"] = round(streaming["value"], 2)
            line["streaming"] = {"value": round(streaming["value"], 2), "unit": "skeletons/s", "fraction_of_resident": round(streaming["value"] / main_res["value_per_gpu"], 4),
                                 "ms_per_step": round(streaming["ms_per_step"], 3), "contexts": streaming["contexts"], "sequences_per_context": streaming["sequences_per_context"],
                                 "h2d_GB_per_s": round(streaming["h2d_gb_per_s"], 2), "d2h_GB_per_s": round(streaming["d2h_gb_per_s"], 2), "host_threads": streaming["host_threads"], "step_ms_context0": streaming["step_ms_context0"],
                                 "pinned_output_numa_nodes_context0": streaming["pinned_output_numa_nodes_context0"], "source_frames_numa_nodes": streaming["source_frames_numa_nodes"], "kernel_numa_balancing": streaming["kernel_numa_balancing"],
                                 "settle_s_before_the_pass": settle, "copy_stream_probe": streaming["copy_stream_probe"], "host_call_ms_context0": streaming["host_call_ms_context0"],
                                 "h2d_MB_per_step": round(streaming["h2d_mb_per_step"], 1), "n_gpus": 1,
                                 "note": "PCIe-inclusive: every step's frames come from host memory (ragged, pinned, double-buffered) and the published "
                                         "persons go back to host memory; never `value`.  At 16cam6p a frame is ~40 KB in and ~20 KB out: PCIe Gen5 x16 "
                                         "bounds this rate, not the kernels (DESIGN 5)"}
        if sec is not None:
            if sec_serial is not None:
                roof2 = roofline_of(sec_serial, WORKLOADS["256cam128p"], True, "serial pass: one context, one stream, %d sequence x %d frames (%.0f skeletons/s alone)"
                                    % (sec_serial["S"], sec_serial["T"], sec_serial["value"]))
                tr2 = load("pmc_traffic_256cam128p_1ctx.json")   # committed PMC passes of `bench.py --workload 256cam128p --contexts 1`
                if tr2:
                    per_launch = tr2["kernels"].get(roof2["kernel"], {}).get("hbm_bytes_per_launch_corrected")
                    # column chunks: the "launch" the HIP events bracket is the pass's whole pair phase = one kernel launch per chunk (16 from 128 cameras)
                    n_launch = max(1, sec_serial["pair_chunks"]) if roof2["kernel"] == "epipolar_pairs" else 1   # what the context actually runs with (se3d_get_pair_chunks)
                    roof2["traffic"] = per_launch * n_launch if per_launch is not None else None
                    roof2["traffic_source"] = "%s: %s bytes per kernel launch x %d launches per pass" % (tr2["_file"], per_launch, n_launch)
            else:
                roof2 = roofline_of(sec, WORKLOADS["256cam128p"], True, "overlapped contexts (%d per GPU)" % sec["NC"])
            roof2["valu"] = valu_of(sec, WORKLOADS["256cam128p"], True, load("pmc_sq_256cam128p_1ctx.json"))
            line["secondary"] = {"workload": "256cam128p, stage 1+2, %d contexts x %d sequences x %d frames per GPU" % (sec["NC"], sec["S"], sec["T"]),
                                 "value": round(sec["value"], 2), "unit": "skeletons/s", "n_gpus": world, "value_per_gpu": round(sec["value_per_gpu"], 2),
                                 "ms_per_step": round(sec["ms_per_step"], 3), "steps": sec["steps"], "warmup": sec["warmup"],
                                 "skeletons_per_step_per_gpu": sec["n_skel_local"], "max_hypotheses_per_frame": sec["max_hyp_per_frame"],
                                 "target": "north_star: >= 1e5 skeletons/s on 8 GPUs = 12 500 per GPU", "exchange_verified": sec["exchange_verified"],
                                 "second_pass_fraction": round(sec["second_pass_fraction"], 6),
                                 "sweep_steps_running_the_hungarian": {"steps": sec["sweep_stats"][0], "hungarian": sec["sweep_stats"][1],
                                                                       "fraction": round(sec["sweep_stats"][1] / max(1, sec["sweep_stats"][0]), 6),
                                                                       "note": "context 0, all passes: camera steps of the sweep (S3D:588-674) that met hypotheses and detections / those "
                                                                               "whose feasibility mask was ambiguous (S3D:628-634: the Hungarian runs on the full cost matrix)"},
                                 "exchange_ms_per_step": sec["exchange_ms_per_step"], "value_no_exchange": None if sec["value_no_exchange"] is None else round(sec["value_no_exchange"], 2),
                                 "roofline": roof2}
        if hyp is not None:
            line["hypothesis_shard"] = {"workload": "ONE 256cam128p stream, %d-frame walks, every rank holds the same frames; hypotheses / detections "
                                                    "split by index %% %d, association + lifecycle replicated" % (hyp["frames_per_step"], world),
                                        "n_gpus": world, "ms_per_frame": round(hyp["ms_per_frame"], 3), "skeletons_per_s": round(hyp["skeletons_per_s"], 1),
                                        "verified_equal_to_single_rank": hyp["verified_equal_to_single_rank"],
                                        "allgather_bytes_per_step": hyp["allgather_bytes_per_step"],
                                        "note": "latency mode of north_star config 5; throughput runs shard sequences (the headline / secondary values)"}
        elif hyp_error is not None:
            line["hypothesis_shard"] = {"error": hyp_error, "n_gpus": world}
        if sec_serial is not None:
            line["secondary"]["serial"] = {"value": round(sec_serial["value"], 2), "ms_per_step": round(sec_serial["ms_per_step"], 3), "steps": sec_serial["steps"],
                                           "contexts": 1, "frames_per_step": sec_serial["B"]}
        if want_parity and hyp_hung:   # the GPU may still be spinning in the collective that never returned: nothing measured after it is trustworthy
            line["parity"] = line["latency_B1"] = {"skipped": "a collective of the hypothesis-shard block did not return within its deadline"}
        elif want_parity:
            line["parity"] = {"16cam6p": bench_parity(headline, rig, seqs_par, 4, local_rank), "64cam32p": bench_parity("64cam32p", rig64, [seq64], 1, local_rank)}
            if want_secondary and not args.no_parity_256:   # north_star's configuration: one frame of the secondary's first sequence, the oracle over the host's cores (~35 s on 256 threads)
                line["parity"]["256cam128p"] = bench_parity("256cam128p", rig2, seqs_sec[0][:1], 1, local_rank, omp_threads=os.cpu_count() or 1, max_steps=1)
            line["parity"].update({
                              "note": "HIP path vs the CPU oracle (restatement of the reference, DESIGN 4) on the bench's own scenes, untimed: stage-1 joints "
                                      "(mpjpe_*), association / hypothesis / person tables identical per frame (assoc_exact_rate), refined + predicted joints "
                                      "of the resident stage-2 walk (refined_*), track ids and LM iteration counts per sequence-step; tolerance of the tests: 1 mm"})
            line["latency_B1"] = bench_latency_b1(rig, seqs_par[0], headline, local_rank)
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(rig, _gen_one((w["n_cam"], w["n_persons"], T_, 1000 * seed_rank)), main_res["params"], stage2)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if hyp_hung:   # a collective of the guarded block never returned: the line is out, leave without the final barrier — with a NON-ZERO
        sys.stdout.flush()   # code, so that the launcher tears the other ranks down and a harness sees the failure
        os._exit(3)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
