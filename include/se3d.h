/*
 * se3d.h — C-ABI of libse3d_hip.so: MI355X-native replacement for the hot path of
 * AIS-Bonn/SmartEdgeSensor3DHumanPose (reference paths are relative to the reference root):
 *
 *   stage 1  skeleton_3d  triangulate_persons()   skeleton_3d/src/skeleton_3d_triang_mult_node.cpp:525-997
 *   stage 2  pose_prior   skeletonCallback() body pose_prior/src/pose_prior_mult_node.cpp:505-921
 *
 * The reference has no FFI; its only external contract is the person_msgs topic surface.  This header is
 * therefore the boundary a maintainer binds at the two internal call sites above:
 * flat POD mirrors of the person_msgs messages, plain pointers and sizes, int status codes, no C++ types,
 * no torch types.  All host buffers are caller-owned; the context owns device memory, its HIP stream and the
 * per-sequence track state.  Calls on one context are serialised by the caller (the reference calls stage 1
 * from exactly one worker thread, S3D:1069, and stage 2 from the single ros::spin thread, PP:947-950).
 *
 * The same POD types are used by the CPU oracle (oracle/se3d_oracle.cpp, test infrastructure only).
 */
#ifndef SE3D_H
#define SE3D_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SE3D_NUM_KEYPOINTS 17        /* NUM_KEYPOINTS, S3D:56,1098 */
#define SE3D_NUM_FUSION_KEYPOINTS 21 /* FUSION_BODY_PARTS::NUM_KEYPOINTS, fusion_body_parts.h:24 */

#define SE3D_POSE_SIMPLE 0 /* pose_method "simple" (S3D:1095, PP:930) */
#define SE3D_POSE_H36M 1   /* pose_method "h36m" */

/* status codes (negative = error); the reference's void functions become int-returning */
#define SE3D_OK 0
#define SE3D_ERR_ARG (-1)      /* null pointer / inconsistent sizes */
#define SE3D_ERR_CAPACITY (-2) /* a frame exceeds the limits given at se3d_create */
#define SE3D_ERR_DEVICE (-3)   /* HIP runtime error (see se3d_last_error) */
#define SE3D_ERR_STATE (-4)    /* call sequence error (e.g. fetch before run) */

/* per-frame flag bits in se3d_frame_out.flags */
#define SE3D_FLAG_HYP_OVERFLOW 1u /* more hypotheses than limits.max_hyp were spawned; result truncated */
#define SE3D_FLAG_FEW_CAMERAS 2u  /* < 2 cameras with detections: empty result (S3D:557-560) */
#define SE3D_FLAG_ASSIGNMENT_FAILED 4u /* the sweep's Hungarian hit its iteration guard (cannot happen with finite costs): that camera
                                          step kept the provisional assignment */

/* person_msgs/Keypoint2D.msg */
typedef struct se3d_keypoint2d {
  float x, y, score;
  float cov[3]; /* xx, xy, yy (pixel^2) */
} se3d_keypoint2d;

/* person_msgs/Person2D.msg with keypoints[] fixed to the 17 the nodes require (S3D:319-320) */
typedef struct se3d_person2d {
  float score;
  float bbox[4];
  se3d_keypoint2d keypoints[SE3D_NUM_KEYPOINTS];
} se3d_person2d;

/* person_msgs/KeypointWithCovariance.msg */
typedef struct se3d_keypoint3d {
  double joint[3]; /* geometry_msgs/Point */
  float score;
  float _pad;
  double cov[6]; /* xx, xy, xz, yy, yz, zz */
} se3d_keypoint3d;

/* person_msgs/PersonCov.msg; bbox_center/bbox_size are never written by either node and are omitted */
typedef struct se3d_person3d {
  uint32_t id; /* stage 1 leaves 0; stage 2 writes the track id (PP:595-596) */
  float score; /* left 0 by both stages */
  se3d_keypoint3d keypoints[SE3D_NUM_FUSION_KEYPOINTS];
} se3d_person3d;

/* Static per-rig tensors produced once by skeleton_3d main() (S3D:1184-1214). */
typedef struct se3d_rig {
  int32_t n_cam;
  const float *Rt; /* [n_cam][12] row-major 3x4 base->camera extrinsics (S3D:1208-1211) */
  const float *K;  /* [n_cam][4]  fx, fy, cx, cy (S3D:314-317) */
  const float *F;  /* [n_cam*(n_cam-1)/2][9] row-major, pair order of get_fundamental_idx (S3D:242-253) */
} se3d_rig;

/* ROS private parameters of both nodes (S3D:1095-1097, PP:930-932). */
typedef struct se3d_params {
  int32_t pose_method;  /* SE3D_POSE_SIMPLE | SE3D_POSE_H36M */
  int32_t norm_height;  /* pose_prior ~norm_height */
  double max_epi_dist;  /* skeleton_3d ~max_epi_dist (default 0.050, demo launch 0.045) */
} se3d_params;

/* Capacities fixed at create time: no allocation per frame after warm-up. */
typedef struct se3d_limits {
  int32_t max_frames;      /* frames per batch (B) */
  int32_t max_det_per_cam; /* detections per camera per frame (P_max) */
  int32_t max_hyp;         /* person hypotheses per frame (H_max >= P_max) */
  int32_t max_tracks;      /* stage-2 tracks per sequence */
} se3d_limits;

/* Options of a context (se3d_create_ex, se3d_set_options); all zero = the library's defaults = what se3d_create uses.  The library reads NO
 * environment variable.  First group: what a deployment tunes (the reference has no counterpart: its nodes are single-threaded CPU code).  Second
 * group: kernel FORMS kept in the library because a parity test runs them against each other and against the oracle — same results (association
 * indices identical, numbers within the tolerances of DESIGN.md section 4); forms that lost their measurement and are not such a reference were
 * removed in round 5 (profiles/r5/README.md). */
typedef struct se3d_options {
  /* --- deployment (the first three are fixed at creation: se3d_set_options rejects a change) */
  int32_t pair_chunks;        /* camera-column chunks of the pair-cost table on large rigs: 0 = default (16 from 128 cameras, 8 from 32 with more than 16
                                 detections per camera, else the whole table), -1 = always the whole table, m >= 2 = m chunks */
  int32_t hi_priority_stream; /* second, high-priority HIP stream for the launch chains (per-camera sweep, per-step stage 2) of contexts with
                                 max_hyp > 64: 0 = default (on), -1 = off — recommended for more than four large-rig contexts per GPU */
  int32_t reserve_cus;        /* CUs the bulk stream leaves to the high-priority stream: 0 = default (32 from 128 cameras when the whole pair table is
                                 materialised, else none), -1 = none, R = R CUs */
  int32_t graph;              /* one-frame calls of a max_frames = 1 context replay one hipGraph each: 0 = default (on), -1 = direct launches */
  int32_t walk_shape;         /* fused stage-2 walk: 0 = automatic (one wavefront per sequence from 256 sequences per launch at max_hyp <= 16),
                                 SE3D_WALK_WAVE, SE3D_WALK_WG */
  int32_t sweep_lds_kb;       /* chunk mode: LDS (KB, <= 128) the per-camera Munkres may use for its working matrix; 0 = L2 (default) */
  /* --- kernel forms (parity tests; 0 = the library's choice) */
  int32_t pairs_form;         /* SE3D_PAIRS_AUTO | _FLAT (stand-alone normalisation + one lane per table entry with IEEE divide: the reference form) |
                                 _ROWS (the large-rig kernel on any rig) | _FUSED (one workgroup per frame even for small batches) */
  int32_t sweep_form;         /* SE3D_SWEEP_AUTO | _PER_CAMERA (cost + assignment launch per camera) | _WAVE_GLOBAL (one wavefront per frame, state in L2) */
  int32_t tri_form;           /* SE3D_TRI_AUTO | _JOINT (one wavefront per (hypothesis, joint) on small rigs too) */
  int32_t post_form;          /* SE3D_POST_AUTO | _BLOCK (the thread-per-hypothesis kernel on small rigs too) */
  int32_t lm_form;            /* normal equations of the per-step LM kernel: SE3D_LM_TREE (default) | _TREE_MFMA | _DENSE (fp64 MFMA J^T J) */
  int32_t walk_form;          /* se3d_run_stage2_walk: SE3D_WALK_FUSED (default) | _STEPS (the per-step kernels back to back) */
  int32_t walk_lm;            /* LM inside the fused walk: SE3D_WALK_LM_TRIO (three persons per wavefront, default) | _SOLO */
  int32_t walk_book;          /* one-wavefront walk's per-step bookkeeping: 0 = LDS (default), 1 = global memory */
  int32_t refine_steps;       /* se3d_refine on a one-sequence context: 0 = one fused-walk launch (default), 1 = the five per-step launches */
  int32_t munkres_form;       /* se3d_hungarian: SE3D_MUNKRES_AUTO | _BLOCK | _WAVE */
  int32_t reproj_form;        /* feedback reprojection: 0 = per-camera lists written directly where they fit LDS (default), 1 = dense + scan + compaction */
  int32_t stream_out;         /* se3d_stream_submit / _collect: 0 = device buffers + D2H copies on the device's shared download stream (default), 1 = the
                                 compaction kernel writes the published persons straight into the pinned host buffers (no copy engine; measured slower in
                                 round 5: 8.3 M against 10 - 11 M skeletons/s, profiles/r5/README.md section 5) */
  int32_t exact_cov;          /* covariances of the joints the triangulation's float32 second pass redoes: SE3D_EXACT_COV_AUTO (default: float32 sigma
                                 samples only where the kept views' systems are ill conditioned or the covariance is wide, the fast path's fp64 samples
                                 around the float32 point elsewhere — se3d_get_exact_list marks those entries) | SE3D_EXACT_COV_F32 (float32 samples for
                                 every listed joint: the oracle's covariance bits on all of them, the form of rounds 4 - 5) */
  int32_t reserved[5];        /* must be 0 */
} se3d_options;
#define SE3D_EXACT_COV_AUTO 0
#define SE3D_EXACT_COV_F32 1
#define SE3D_WALK_WAVE 1
#define SE3D_WALK_WG 2
#define SE3D_PAIRS_AUTO 0
#define SE3D_PAIRS_FLAT 1
#define SE3D_PAIRS_ROWS 2
#define SE3D_PAIRS_FUSED 3
#define SE3D_SWEEP_AUTO 0
#define SE3D_SWEEP_PER_CAMERA 1
#define SE3D_SWEEP_WAVE_GLOBAL 2
#define SE3D_TRI_AUTO 0
#define SE3D_TRI_JOINT 1
#define SE3D_POST_AUTO 0
#define SE3D_POST_BLOCK 1
#define SE3D_LM_TREE 0
#define SE3D_LM_TREE_MFMA 1
#define SE3D_LM_DENSE 2
#define SE3D_WALK_FUSED 0
#define SE3D_WALK_STEPS 1
#define SE3D_WALK_LM_TRIO 0
#define SE3D_WALK_LM_SOLO 1
#define SE3D_MUNKRES_AUTO 0
#define SE3D_MUNKRES_BLOCK 1
#define SE3D_MUNKRES_WAVE 2

/* One synchronised set of per-camera Person2DList messages (input of triangulate_persons, S3D:525). */
typedef struct se3d_frame_in {
  int32_t n_cam;
  const int32_t *det_offset;    /* [n_cam+1] prefix offsets into persons (camera i has det_offset[i+1]-det_offset[i]) */
  const se3d_person2d *persons; /* [det_offset[n_cam]] */
  const float *fb_delay;        /* [n_cam] Person2DList.fb_delay, may be NULL (= -1) */
  double stamp;                 /* header.stamp of the pivot message, seconds (S3D:1060) */
} se3d_frame_in;

/* PersonCovList payload produced by stage 1 (S3D:1059-1071) plus the association table. */
typedef struct se3d_frame_out {
  int32_t capacity;       /* in : persons[]/person_hyp[] hold this many entries */
  int32_t hyp_capacity;   /* in : hyp_assoc holds this many rows (0 = not wanted) */
  int32_t n_persons;      /* out: published persons after duplicate merge (S3D:984-996) */
  int32_t n_hyp;          /* out: hypotheses created by the greedy sweep (S3D:565-674), creation order */
  uint32_t flags;         /* out: SE3D_FLAG_* */
  se3d_person3d *persons; /* out: [capacity], canonical order = ascending hypothesis index (serial reference) */
  int32_t *person_hyp;    /* out: [capacity] hypothesis index each person came from (lower index after a merge) */
  int32_t *hyp_assoc;     /* out: [hyp_capacity][n_cam] index into that camera's persons[] or -1 */
} se3d_frame_out;

/* persons3d_fused / persons3d_fused_pred payload produced by stage 2 (PP:905-907). */
typedef struct se3d_fused_out {
  int32_t capacity;
  int32_t n_persons;         /* persons published (tracks with num_obs > 10, PP:845-848) */
  int32_t n_tracks;          /* live tracks after removal/merge */
  float pred_delta_t;        /* value written into fb_delay_per_cam (PP:526-531) */
  se3d_person3d *fused;      /* [capacity] */
  se3d_person3d *fused_pred; /* [capacity] */
  int32_t *lm_iterations;    /* [capacity] outer LM iterations used for each published person (diagnostic) */
  uint32_t flags;            /* out: SE3D_S2FLAG_* of the sequence, sticky since se3d_refine_reset */
  double *lm_objective;      /* [capacity] or NULL (diagnostic, round 5): the factor graph's error at the published values, GTSAM's
                                graph.error(result) = 0.5 sum ||whitened residual||^2 over the unary, root and range factors (PP:626-758), in the
                                normalised coordinates the optimiser works in; -1 for a person whose optimisation failed (published at its
                                measurements, PP:751-758).  A solver-independent yardstick: two LM implementations that stop at different points
                                of a badly scaled problem can be compared by it. */
} se3d_fused_out;

/* per-sequence flag bits in se3d_fused_out.flags */
#define SE3D_S2FLAG_TRACK_OVERFLOW 1u    /* more live tracks than limits.max_tracks: a detection got no track */
#define SE3D_S2FLAG_ASSIGNMENT_FAILED 4u /* the track-association Hungarian did not terminate (cannot happen with finite costs) */
#define SE3D_S2FLAG_NOT_A_FOREST 8u      /* bone factors of a person did not form a forest (impossible with the reference's bone tables):
                                            that person was published at its measurements, as after a failed optimisation (PP:751-758) */

typedef struct se3d_ctx se3d_ctx;

/* --- rig set-up (host; replaces S3D:1187-1211) ---------------------------------------------------------
 * T_base_to_cam: [n_cam][16] row-major 4x4 doubles as returned by tf lookupTransform(cam, base).
 * Writes Rt_out [n_cam][12] floats and F_out [n_cam*(n_cam-1)/2][9] floats (either may be NULL). */
int se3d_rig_from_transforms(int32_t n_cam, const double *T_base_to_cam, float *Rt_out, float *F_out);

/* --- context ------------------------------------------------------------------------------------------- */
/* Stream semantics: a context works on its own non-blocking HIP stream (se3d_stream_handle).  One exception: with options.reserve_cus > 0 (or
 * a rig of >= 128 cameras run with options.pair_chunks = -1) the stream is created by hipExtStreamCreateWithCUMask, which yields a BLOCKING stream:
 * work on the legacy null stream (plain hipMemcpy, a torch op on the default stream) then synchronises with it — keep such work out of
 * timed regions. */
se3d_ctx *se3d_create(const se3d_rig *rig, const se3d_params *params, const se3d_limits *limits, int32_t device_id);
/* As se3d_create with options (NULL = defaults). */
se3d_ctx *se3d_create_ex(const se3d_rig *rig, const se3d_params *params, const se3d_limits *limits, const se3d_options *options, int32_t device_id);
/* Replaces the options of a live context (the creation-time fields pair_chunks, hi_priority_stream, reserve_cus must keep their values:
 * SE3D_ERR_ARG otherwise; so must a non-zero reserved word).  Takes effect from the next call on. */
int se3d_set_options(se3d_ctx *ctx, const se3d_options *options);
int se3d_get_options(se3d_ctx *ctx, se3d_options *options_out);
void se3d_destroy(se3d_ctx *ctx);
/* ctx != NULL: message of the context's last failed call.  ctx == NULL: why the last se3d_create of the calling thread returned NULL
 * (unsupported sizes, no device, out of device memory — with the bytes the context needs and the bytes free). */
const char *se3d_last_error(const se3d_ctx *ctx);
/* Device memory in bytes that se3d_create allocates for a rig of n_cam cameras under `limits` (host arithmetic only, no device call; the
 * streaming, reprojection and hypothesis-shard buffers of later calls are not included); -1 for sizes se3d_create rejects.  bench.py's
 * memory plan (N contexts per GPU) and a node's start-up check use it to fail with a message instead of a hipMalloc error. */
int64_t se3d_device_bytes(int32_t n_cam, const se3d_limits *limits);
int64_t se3d_device_bytes_ex(int32_t n_cam, const se3d_limits *limits, const se3d_options *options);
const char *se3d_version(void);
/* Layout version of the structs of this header: bumped whenever a struct gains, loses or moves a field (6: se3d_options.exact_cov; 5:
 * se3d_fused_out.lm_objective).  The out-structs carry no size field, so a caller compiled against another layout would make the library
 * read or write past its structs: compare se3d_abi_version() of the loaded library with the SE3D_ABI_VERSION the caller was compiled with
 * before the first call (host/se3d_nodes.cpp and the Python mirror refuse a mismatch). */
#define SE3D_ABI_VERSION 6
int32_t se3d_abi_version(void);

/* --- stage 1: drop-in for triangulate_persons (S3D:525-997) ------------------------------------------- */
/* one frame, synchronous (host in, host out).  On a context with limits.max_frames == 1 — one synchronised frame per callback, the reference's
 * operating point (S3D:1069) — the call replays ONE hipGraph (upload, kernel chain, download) captured on first use; the same holds for
 * se3d_refine on such a context (PP:505).  options.graph = -1 or se3d_profile_enable(ctx, 1) select direct launches. */
int se3d_triangulate(se3d_ctx *ctx, const se3d_frame_in *frame, se3d_frame_out *out);
/* B frames, synchronous */
int se3d_triangulate_batch(se3d_ctx *ctx, int32_t n_frames, const se3d_frame_in *frames, se3d_frame_out *outs);

/* resident pipeline for throughput: stage (H2D) once, run many times with inputs in HBM, fetch (D2H) */
int se3d_stage_frames(se3d_ctx *ctx, int32_t n_frames, const se3d_frame_in *frames);
int se3d_run_stage1(se3d_ctx *ctx); /* asynchronous on the context stream */
/* The same pass in parts, for callers that schedule several contexts on one GPU (bench.py: at 256 cameras the VALU-bound pair-cost
 * kernels of the contexts are chained by events so that one runs at a time while the latency-bound camera sweeps and stage-2 steps of
 * the other contexts fill the gaps).  phases = bit mask, executed in this order: 1 normalisation + pair-cost table (rows N, E),
 * 8 camera sweep (rows G, M), 2 triangulation (rows T, P, O, U), 4 post-processing + duplicate merge (rows X, D); 15 = se3d_run_stage1. */
int se3d_run_stage1_phases(se3d_ctx *ctx, int32_t phases);
int se3d_sync(se3d_ctx *ctx);
/* The context's HIP stream (hipStream_t) so that a host framework can order its own work (e.g. an RCCL collective on the packed
 * skeletons) after the asynchronous calls of this context without a host-side synchronisation. */
void *se3d_stream_handle(se3d_ctx *ctx);
int se3d_fetch_stage1(se3d_ctx *ctx, int32_t n_frames, se3d_frame_out *outs);

/* --- config 2 of the benchmark: batched DLT triangulation with the association GIVEN -----------------
 * Rows N (S3D:312-333), T (:440-465), P (:425-438), O (:748-844) and U (:471-523) per hypothesis and joint, scattered into the
 * 21-slot fusion layout (:849-857): what the reference's per-hypothesis loop does before its plausibility filters, with the
 * greedy sweep replaced by the caller's table.  assoc: [n_frames][n_hyp][n_cam] index into that camera's persons[] of the frame or
 * -1 (the layout of se3d_frame_out.hyp_assoc); persons_out: [n_frames][n_hyp]; a joint with < 2 usable views and a hypothesis with
 * < 2 views are default-constructed (score 0).  No post-processing (X), no merge (D). */
int se3d_triangulate_assoc(se3d_ctx *ctx, int32_t n_frames, const se3d_frame_in *frames, int32_t n_hyp, const int32_t *assoc,
                           se3d_person3d *persons_out);
/* resident form: se3d_stage_frames, then se3d_stage_assoc once; se3d_run_dlt is asynchronous on the context stream */
int se3d_stage_assoc(se3d_ctx *ctx, int32_t n_hyp, const int32_t *assoc);
int se3d_run_dlt(se3d_ctx *ctx);
int se3d_fetch_dlt(se3d_ctx *ctx, int32_t n_frames, se3d_person3d *persons_out);

/* --- stage 2: drop-in for the pose_prior skeletonCallback body (PP:505-921) ----------------------------
 * The context holds n_seq independent sequences (each the global state of one pose_prior node: g_tracks,
 * g_fb_delay_buffer, g_frame_nr, g_t_prev, g_next_id; PP:54-60,123).  Sequence s consumes, per call, frame
 * slot s of the most recent stage-1 run (resident path) or host measurements (se3d_refine). */
int se3d_refine_reset(se3d_ctx *ctx);
/* The state of one pose_prior instance (its g_tracks in vector order and the node globals) as a host blob that does not depend on the
 * context's limits: se3d_refine_save from one context, se3d_refine_load into another (e.g. one created with larger limits — the
 * reference's g_tracks grows without bound, PP:570-580; host/se3d_nodes.cpp grows its context this way instead of truncating).
 * se3d_refine_state_bytes: upper bound of a blob of this context.  Loading re-numbers the track slots (track k of the vector -> slot k);
 * results do not depend on the slot numbering.  A blob is deterministic: joints outside a track's previous estimate are written as zeros,
 * and save -> load -> save gives the same bytes (tests/test_gpu_stage2.py).  Synchronous. */
int64_t se3d_refine_state_bytes(const se3d_ctx *ctx);
int se3d_refine_save(se3d_ctx *ctx, int32_t seq, void *blob, int64_t capacity_bytes, int64_t *bytes);
int se3d_refine_load(se3d_ctx *ctx, int32_t seq, const void *blob, int64_t bytes);
/* one PersonCovList in, fused lists out; seq selects the pose_prior instance */
int se3d_refine(se3d_ctx *ctx, int32_t seq, int32_t n_det, const se3d_person3d *dets, int32_t n_cam,
                const float *fb_delay_per_cam, double t, se3d_fused_out *out);
/* resident: sequence s < n_seq consumes frame slot frame_slot0 + s of the last se3d_run_stage1, stamp t[s]
 * (a batch staged as [time step k][sequence s] is walked with frame_slot0 = k * n_seq, one call per time step) */
int se3d_run_stage2(se3d_ctx *ctx, int32_t n_seq, int32_t frame_slot0, const double *t);
int se3d_fetch_stage2(se3d_ctx *ctx, int32_t n_seq, se3d_fused_out *outs);
/* Multi-step replay of resident sequences: equivalent to n_steps consecutive se3d_run_stage2 calls, step k consuming frame slots
 * frame_slot0[k] .. frame_slot0[k]+n_seq-1 at times t[k*n_seq + s] (n_steps <= 64, n_steps*n_seq <= limits.max_frames).  When the
 * association problem is small the whole walk is ONE kernel launch (a workgroup per sequence loops over the time steps), which
 * removes the per-step launch and tail latency; otherwise the per-step kernels are launched back to back.  Results are kept per
 * frame slot: se3d_fetch_stage2_walk(frame_slot0, n_seq) returns the persons3d_fused / _pred lists of the step that consumed those
 * slots; se3d_pack_stage2_walk_device copies [n_frames][max_hyp] persons + [n_frames] counts (frame-slot order, asynchronous). */
int se3d_run_stage2_walk(se3d_ctx *ctx, int32_t n_seq, int32_t n_steps, const int32_t *frame_slot0, const double *t);
int se3d_fetch_stage2_walk(se3d_ctx *ctx, int32_t frame_slot0, int32_t n_seq, se3d_fused_out *outs);
int se3d_pack_stage2_walk_device(se3d_ctx *ctx, int32_t n_frames, void *d_persons, void *d_counts);
/* Same with a caller-chosen number of person slots per frame (1 <= slots <= max_hyp): [n_frames][slots] persons.  Lets an exchange
 * carry only as many slots as the scene can publish instead of the full hypothesis capacity; frames that published more than
 * `slots` persons are truncated in the copy (their count is still the true one, so the receiver can detect it). */
int se3d_pack_stage2_walk_device_slots(se3d_ctx *ctx, int32_t n_frames, int32_t slots, void *d_persons, void *d_counts);
/* Dense form for an exchange that should move skeletons, not padding: the published fused persons of frame slots 0 .. n_frames-1 back to
 * back in frame-slot order into d_persons [capacity] and d_offsets [n_frames + 2] (int32): d_offsets[f] .. d_offsets[f + 1] is the range of
 * frame slot f, d_offsets[n_frames] the number of persons written and d_offsets[n_frames + 1] the number that did not fit `capacity`
 * (they are dropped from the copy, never silently: the receiver reads this word).  Asynchronous on the context's stream. */
int se3d_pack_stage2_walk_device_dense(se3d_ctx *ctx, int32_t n_frames, int32_t capacity, void *d_persons, void *d_offsets);


/* --- streaming ingest: frames arrive from the host every step (the reference's situation at S3D:1069) ---------------------------
 * se3d_stream_submit takes one batch of frames ([time step k][sequence s] order, frame slot = k * n_seq + s as in the resident path),
 * copies the callers' ragged Person2D payloads WITHOUT padding into pinned memory, and enqueues, without waiting: H2D on a copy stream,
 * the expansion into the resident layout, stage 1, the stage-2 walk over n_steps time steps (arguments as se3d_run_stage2_walk), a
 * device-side compaction of the PUBLISHED persons and their D2H on a third stream.  Two batches may be in flight: the copies of batch
 * k + 1 and k - 1 overlap the kernels of batch k.  se3d_stream_collect waits for the OLDEST batch and hands out its results in pinned
 * host memory owned by the context (valid until the next-but-one submit): persons of frame slot i are
 * fused[person_offset[i] .. person_offset[i + 1]).  Do not mix with the resident calls (se3d_stage_frames ...) on one context. */
typedef struct se3d_stream_result {
  int32_t n_frames;
  int32_t n_persons;               /* published persons in the batch = person_offset[n_frames] */
  const int32_t *person_offset;    /* [n_frames + 1] */
  const se3d_person3d *fused;      /* [n_persons] persons3d_fused payloads, frame-slot order */
  const se3d_person3d *fused_pred; /* [n_persons] persons3d_fused_pred */
} se3d_stream_result;
int se3d_stream_submit(se3d_ctx *ctx, int32_t n_frames, const se3d_frame_in *frames, int32_t n_seq, int32_t n_steps, const int32_t *frame_slot0,
                       const double *t);
int se3d_stream_collect(se3d_ctx *ctx, se3d_stream_result *out);
/* The streaming entry points move every context's uploads over ONE stream per device and every download over another (copies of one direction on
 * several streams interleave on the copy engines: 37 - 43 GB/s instead of 56).  The pair is created by the first se3d_stream_submit on the device and
 * lives as long as the process (at most 64 devices; a hipDeviceReset invalidates it).  Its health is WATCHED (round 6): a burst of downloads while
 * the upload stream is idle can leave both directions on one copy engine for good — the streaming rate then drops to 60 % until the streams are
 * re-created, which round 5 saw in one run of ten.  se3d_stream_collect enqueues a 1 MB companion upload next to every batch's downloads (a download
 * that never starts with the upload stream idle does not trigger it), and every download of >= 8 MB is timed with events on the download stream: six in
 * a row below 30 GB/s replace the pair by a new one (pending copies of the old pair finish; at most 16 repairs per device and process).
 * Diagnostic: *gbps_both = GB/s per direction of four 32 MB uploads and four 32 MB downloads queued at once on the device's current pair (~48
 * healthy, ~28 serialised; ~10 ms, 2 x 32 MB of pinned and of device memory for its duration; only call it with no batch in flight), *repairs
 * (optional) = how often the pair was replaced in this process, *last_download_gbps (optional) = rate of the last timed download. */
int se3d_copy_stream_probe(int32_t device, double *gbps_both, int32_t *repairs, double *last_download_gbps);
/* --- feedback reprojection: drop-in for fusedSkeletonCallback (pose_reprojection/src/skeleton_reproj_mult_node.cpp:139-235) --
 * Projects fused/predicted 3D skeletons and their covariances (7-sigma-point unscented transform, :62-75) into every camera and
 * returns one Person2DList payload per camera (topic <cam>/skel_pred).  Cameras in double precision as the node uses them:
 * T_base_to_cam [n_cam][16] row-major 4x4 (tf lookupTransform(cam, base), :82), P [n_cam][4] = fx, fy, cx, cy of the CameraInfo
 * projection matrix, wh [n_cam][2] = image width, height (:161-162, :207).  Call se3d_set_reprojection_cameras once. */
typedef struct se3d_reproj_out {
  int32_t capacity;          /* in : persons[] holds this many entries */
  int32_t *det_offset;       /* out: [n_cam+1] prefix offsets: camera i received persons[det_offset[i] .. det_offset[i+1]) */
  se3d_person2d *persons;    /* out: projected persons (score 1.0, bbox = extent of the valid keypoints, :227) */
  int32_t *person_index;     /* out: [capacity] index into the input persons of each projected person (may be NULL) */
} se3d_reproj_out;
int se3d_set_reprojection_cameras(se3d_ctx *ctx, const double *T_base_to_cam, const double *P_fx_fy_cx_cy, const int32_t *wh);
int se3d_reproject(se3d_ctx *ctx, int32_t n_persons, const se3d_person3d *persons, se3d_reproj_out *out);
/* Same, on the predicted skeletons (persons3d_fused_pred) that stage 2 left in HBM: no host round trip between pose_prior and the
 * reprojection.  `slot` selects the result: the sequence index after se3d_run_stage2 / se3d_refine, the FRAME slot (frame_slot0[k] + s)
 * after se3d_run_stage2_walk, whose results are kept per frame slot.  out == NULL: launch only (asynchronous). */
int se3d_reproject_resident(se3d_ctx *ctx, int32_t slot, se3d_reproj_out *out);
/* The reprojection of a replayed batch: result slots slot0 .. slot0 + n_slots - 1 (frame slots after se3d_run_stage2_walk) in ONE launch —
 * n_slots x fusedSkeletonCallback (REPROJ:139-235).  Asynchronous; the per-camera lists stay on the device, se3d_fetch_reprojection copies
 * the lists of one slot of the last batch out (same layout as se3d_reproject). */
int se3d_reproject_resident_batch(se3d_ctx *ctx, int32_t slot0, int32_t n_slots);
int se3d_fetch_reprojection(se3d_ctx *ctx, int32_t slot, se3d_reproj_out *out);

/* --- row M as a stand-alone entry: drop-in for HungarianAlgorithm::assignmentoptimal (Hungarian.cpp:60) on the GPU.
 * dist is column-major [row + n_rows*col] (host memory), n_rows, n_cols <= 256; assignment[row] = col or -1. */
int se3d_hungarian(se3d_ctx *ctx, int32_t *assignment, double *cost, const double *dist, int32_t n_rows, int32_t n_cols);

/* --- diagnostic accessor: pairwise epipolar cost block of camera pair (cam_a < cam_b) of staged frame `frame` after
 * se3d_run_stage1: out[det_a * max_det_per_cam + det_b] = mean symmetric epipolar distance (calcCost inner loop,
 * S3D:353-368) in FILTERED detection slots, -1 where no joint is shared; n_a/n_b receive the valid slot counts. */
int se3d_get_pair_costs(se3d_ctx *ctx, int32_t frame, int32_t cam_a, int32_t cam_b, double *out, int32_t *n_a, int32_t *n_b);
/* Large rigs keep camera-column chunks of the pair table instead of the whole (se3d_get_pair_chunks > 0): there se3d_get_pair_costs is not a
 * plain read — it RECOMPUTES the block with the pass's kernel into scratch of the context (pair slot 0 of the frame, first element of the cost
 * matrix), which the next se3d_run_stage1 overwrites anyway.  In every mode entries outside [0, n_a) x [0, n_b) are undefined. */
int se3d_get_pair_chunks(se3d_ctx *ctx);

/* --- diagnostic accessor: the FUSION_BODY_PARTS constants compiled into the kernels (fusion_body_parts.h:3-24, 33): indices22 = the joint
 * indices in the enum's declaration order, NUM_KEYPOINTS last (-1 for joints the kernels never name), vel_sigmas21 read back from the
 * device's constant table. */
int se3d_get_fusion_constants(se3d_ctx *ctx, int32_t *indices22, double *vel_sigmas21);

/* --- diagnostic self-test: the pair-cost kernels form sqrtf(x) and a / b of calcCost (S3D:358-361) with the arithmetic cores of the IEEE
 * sequences (no operand scaling / fix-up) inside [2^-60, 2^60]; this runs both against the device's IEEE results — the square root on every
 * float of the range, the quotient on n_div pseudo-random operand pairs — and returns the mismatch counts (expected: 0, 0). */
int se3d_selftest_exact_math(se3d_ctx *ctx, uint64_t n_div, uint64_t *sqrt_mismatches, uint64_t *div_mismatches);

/* --- diagnostic (round 5): running totals of the per-camera sweep launches of this context (rigs too large for the one-wavefront sweep) —
 * camera steps that met existing hypotheses AND detections (S3D:588-626), and those in which a row or column of the feasibility mask counted more
 * than one entry, i.e. in which the reference runs the Hungarian on the unmasked cost matrix (S3D:628-634).  32-bit counters on the device. */
int se3d_get_sweep_stats(se3d_ctx *ctx, uint64_t *steps, uint64_t *hungarian_steps);

/* --- diagnostic self-test (round 5): the large-rig pair kernel (calcCost's two quotients per joint, S3D:358-361) divides by a line norm through
 * its correctly rounded reciprocal y = RN(1 / b), staged once per (detection, joint), in three instructions: q0 = RN(a y), r = fma(-b, q0, a),
 * q = fma(r, y, q0) (one Markstein correction; csrc/se3d_stage1.hip has the argument why one suffices).  This runs it against the device's
 * IEEE quotient on n pseudo-random operand pairs drawn from the kernel's guarded operand range and returns the mismatch count (expected 0) and
 * the pairs skipped because the divisor fails the guard (significand all ones: such blocks take the IEEE divide in the kernel). */
int se3d_selftest_shared_quotient(se3d_ctx *ctx, uint64_t n, uint64_t *mismatches, uint64_t *skipped);

/* --- diagnostic self-test (round 4): the float32 DLT solve of the triangulation's second pass — triangulate() + calcReprojectionError()
 * (S3D:425-465) with the oracle's restatement of the reference's arithmetic: Eigen 3.3 JacobiSVD of the float rows (scaling, ColPivHouseholderQR preconditioner unless
 * the system is 4 x 4, two-sided Jacobi sweeps), every sum sequential, IEEE division / square root, no FMA — on n_sys caller systems of
 * n_views views each: Rt [n_sys][n_views][12] camera matrices, pts [n_sys][n_views][3] (x, y, conf) in normalised image coordinates.
 * X_out [n_sys][3], err_out [n_sys] (confidence-weighted reprojection error).  Bit-identical to the same sequence on any IEEE-754 host. */
int se3d_selftest_dlt32(se3d_ctx *ctx, int32_t n_sys, int32_t n_views, const float *Rt, const float *pts, int32_t weight_by_conf, float *X_out, double *err_out);

/* --- diagnostic accessor (round 4): the joints the last triangulation pass of this context (se3d_run_stage1 / _phases / se3d_triangulate* /
 * se3d_run_dlt) handed to the float32-exact second pass: entries[i] = (frame * max_hyp + hypothesis) << 5 | input joint index (order not
 * defined; bit 31 set (round 6): the joint's covariance came from fp64 sigma samples around the float32 point, options.exact_cov),
 * *n_entries = their number (may exceed capacity), *total (optional) = running total since se3d_create. */
int se3d_get_exact_list(se3d_ctx *ctx, uint32_t *entries, int32_t capacity, int32_t *n_entries, int64_t *total);

/* --- device-side result access for multi-GPU exchange (RCCL all-gather of final skeletons) -------------
 * Enqueue, on the context stream, device-to-device copies of the results of the last run into caller-provided DEVICE
 * memory (e.g. torch tensors): persons as [n][max_hyp] se3d_person3d (valid entries first) and counts as [n] int32,
 * n = staged frames (stage 1) or n_seq sequences (stage 2, results of the last se3d_run_stage2 call).
 * ASYNCHRONOUS: call se3d_sync before another stream or the host touches the buffers. */
int se3d_pack_stage1_device(se3d_ctx *ctx, void *d_persons, void *d_counts);
int se3d_pack_stage2_device(se3d_ctx *ctx, int32_t n_seq, void *d_persons, void *d_counts);

/* --- hypothesis shard of ONE stream over the ranks of a node (the reference's axes are its OpenMP loops over hypotheses,
 * S3D:676-681, and over detections, PP:582-587).  Every rank stages the same frames; the association (rows N, E, G, M) and the track
 * association / lifecycle (rows A2, L) run on every rank and give identical tables; rank r triangulates the hypotheses h with
 * h % world == r and refines the detections i with i % world == r.  Two exchanges, done by the caller (RCCL all-gather in
 * smartedgesensor3dhumanpose_amd/shard.py) on buffers in device memory:
 *   stage 1: _pre writes [staged frames][ceil(max_hyp / world)] se3d_person3d (own hypotheses; se3d_shard_bytes(ctx, 1, frames));
 *            _post takes the gathered [world][frames][ceil(max_hyp / world)] blocks and runs post-processing + duplicate merge;
 *   stage 2: _pre writes [n_seq][ceil(max_hyp / world)] records (fused + predicted person and the updated track state;
 *            se3d_shard_bytes(ctx, 2, n_seq)); _post applies the gathered [world][n_seq][...] records and runs the lifecycle.
 * All four calls are asynchronous on the context stream; results are fetched with se3d_fetch_stage1 / se3d_fetch_stage2 and equal
 * the single-rank results bit for bit.  se3d_set_shard(ctx, 0, 1) switches sharding off. */
int se3d_set_shard(se3d_ctx *ctx, int32_t rank, int32_t world);
int64_t se3d_shard_bytes(se3d_ctx *ctx, int32_t stage, int32_t n);
int se3d_run_stage1_shard_pre(se3d_ctx *ctx, void *d_pack);
int se3d_run_stage1_shard_post(se3d_ctx *ctx, const void *d_gathered);
int se3d_run_stage2_shard_pre(se3d_ctx *ctx, int32_t n_seq, int32_t frame_slot0, const double *t, void *d_pack);
int se3d_run_stage2_shard_post(se3d_ctx *ctx, int32_t n_seq, const void *d_gathered);

/* --- measurement: HIP-event timing of the kernels on the context stream -------------------------------- */
int se3d_profile_enable(se3d_ctx *ctx, int32_t on);
int se3d_profile_reset(se3d_ctx *ctx);
/* name in {"normalize","epipolar_pairs","sweep","triangulate","postprocess","track_assoc","lm_refine",
 *          "track_update","stage2_walk","dlt"}; returns accumulated ms and launch count ("sweep" = all cost+assign launches of a run,
 *          "dlt" = one se3d_run_dlt pass) */
int se3d_profile_get(se3d_ctx *ctx, const char *name, double *ms_total, int64_t *launches);

#ifdef __cplusplus
}
#endif
#endif /* SE3D_H */
