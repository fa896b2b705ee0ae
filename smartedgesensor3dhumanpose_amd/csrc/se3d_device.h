// se3d_device.h — device-side data layout of libse3d_hip.so (MI355X / gfx950 only).
//
// HBM layout (B = frames per batch, N = cameras, P = max detections per camera, H = max hypotheses):
//   raw      [B][N][P] se3d_person2d      caller's Person2D payloads, dense-padded            (428 B each)
//   kp       [B][N][P][17] float4         normalised (x, y, conf, 0); filtered slot order     (row N; stand-alone pair kernels only —
//                                         small rigs keep the normalised points in LDS, the triangulation normalises in registers)
//   vmask    [B][N][P] uint32 x2          bit j: conf > 0.30 (calcCost gate) / conf >= 0.30 (triangulation gate)
//   pair     [B][N(N-1)/2][P][P] double   pairwise mean symmetric epipolar distance, -1 = no joint in common  (row E); large rigs keep
//                                         two camera-column chunks of it per frame instead of the whole table (Se3dPairPlan)
//   assoc    [B][H][N] int32              filtered detection slot of hypothesis h in camera c, or -1           (row G)
//   C, mask  [B][H*P] double / uint8      per-step cost matrix, column-major with leading dimension H
//   hypP     [B][H] se3d_person3d         per-hypothesis triangulated person                                  (rows T,U,X)
//   out      [B][H] se3d_person3d + counts, person_hyp, hyp_assoc (original detection indices)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/se3d.h"

#define SE3D_NKP 17
#define SE3D_NFU 21
#define SE3D_MAX_CAMS 256   // view lists are staged in LDS per wavefront
#define SE3D_MUNKRES_MAX 256  // rows/cols bit-mask width of the device Munkres (4 x u64)

struct Se3dDims {
  int B, N, P, H, npairs;
  int pose_method, norm_height;
  double max_epi;
  int shard_rank, shard_world;  // hypothesis shard (se3d_set_shard): this context triangulates hypotheses h with h % world == rank
  int pair_stride;              // pair slots per frame of `pair`: npairs (whole table) or 2 x the largest column chunk (se3d_pair_plan)
  int pair_chunks;              // 0: the whole pair table is materialised; m >= 2: camera-column chunks (large rigs), see Se3dPairPlan
  int tri_waves;                // wavefronts of se3d_k_triangulate_exact (one lane-interleaved workspace each in tri_ws)
};

// Column chunks of the pair table (large rigs, round 3).  Camera step b of the sweep (S3D:588-674) reads only the blocks (a, b), a < b —
// column b of the upper-triangular table — so the table never has to exist as a whole: the columns 1 .. N-1 are cut into m chunks of about
// equal pair count, the pair kernel fills chunk j (buffer j % 2) while the sweep walks the columns of chunk j - 1, and a frame keeps
// 2 x cap pair slots instead of N (N - 1) / 2 (256 cameras, m = 16: 4 592 instead of 32 640 — 0.6 instead of 4.3 GB per frame at P = 128).
// Pairs are enumerated column-major inside the plan: list index of (a, b) = b (b - 1) / 2 + a.
#define SE3D_PAIR_CHUNKS_MAX 32
struct Se3dPairPlan {
  int m = 0;                                  // chunks (0: whole table)
  int col0[SE3D_PAIR_CHUNKS_MAX + 1] = {};    // chunk j covers the columns col0[j] .. col0[j + 1] - 1; col0[0] = 1, col0[m] = N
  int list0[SE3D_PAIR_CHUNKS_MAX + 1] = {};   // column-major list index of its first pair: col0[j] (col0[j] - 1) / 2
  int cap = 0;                                // largest chunk, in pairs
};
__host__ __device__ inline int se3d_pair_col_index(int a, int b) { return b * (b - 1) / 2 + a; }   // a < b
inline Se3dPairPlan se3d_pair_plan(int N, int m) {
  Se3dPairPlan pl;
  if (m < 2) return pl;
  if (m > SE3D_PAIR_CHUNKS_MAX) m = SE3D_PAIR_CHUNKS_MAX;
  if (m > N - 1) m = N - 1;
  const long long np = (long long)N * (N - 1) / 2;
  int b = 1, cnt = 0;
  for (int j = 0; j < m; ++j) {
    while (b < N && (long long)b * (b - 1) / 2 < np * j / m) ++b;
    if (cnt > 0 && b <= pl.col0[cnt - 1]) b = pl.col0[cnt - 1] + 1;   // every chunk holds at least one column
    if (b >= N) break;                                                 // no column left: fewer chunks than asked for
    pl.col0[cnt++] = b;
  }
  if (cnt < 2) return Se3dPairPlan();
  m = cnt;
  pl.m = m;
  pl.col0[0] = 1;
  pl.col0[m] = N;
  for (int j = 0; j <= m; ++j) pl.list0[j] = pl.col0[j] * (pl.col0[j] - 1) / 2;
  for (int j = 0; j < m; ++j) pl.cap = pl.list0[j + 1] - pl.list0[j] > pl.cap ? pl.list0[j + 1] - pl.list0[j] : pl.cap;
  return pl;
}
// Which rigs run in chunks: those whose sweep is the chain of per-camera launches anyway (se3d_launch_stage1: neither the one-wavefront nor
// the one-wavefront sweep applies) and whose table is big enough to matter.  options.pair_chunks = m overrides (-1: whole table).
// One predicate for the launcher and the chunk default: the one-wavefront sweep with its state in global memory
// (se3d_k_sweep_fused_wave) takes any number of cameras; only its LDS form (se3d_k_sweep_wave_lds) is limited to N <= 64.
inline bool se3d_sweep_wave_global_ok(int P, int H) { return H <= 64 && P <= 64 && H * P <= 2048; }
inline int se3d_pair_chunks_default(int N, int P, int H) {
  const bool wave_sweep = se3d_sweep_wave_global_ok(P, H);
  if (P <= 16 || N < 32 || wave_sweep) return 0;
  return N >= 128 ? 16 : 8;
}
// Chunks of a context: only where the default applies them (the sweep is the per-camera launch chain, the rows kernel fills the table) and no
// kernel form that needs the whole table is forced.
inline int se3d_pair_chunks_of(const se3d_options &o, int N, int P, int H) {
  const int def = se3d_pair_chunks_default(N, P, H);
  if (def == 0 || o.pair_chunks < 0 || o.pairs_form != SE3D_PAIRS_AUTO || o.sweep_form != SE3D_SWEEP_AUTO) return 0;
  return o.pair_chunks >= 2 ? o.pair_chunks : def;
}

struct Se3dDev {
  Se3dDims d;
  // rig
  const float *Rt;  // [N][12]
  const float *K;   // [N][4]
  const float *F;   // [npairs][9]
  // input
  const se3d_person2d *raw;  // [B][N][P]
  const int32_t *nraw;       // [B][N]
  // normalised
  float4 *kp;         // [B][N][P][17] normalised (x, y, conf, 0): written only when a stand-alone pair kernel runs (large P, A/B switches)
  float *pscore;      // [B][N][P]
  uint32_t *vmask_gt; // [B][N][P]
  uint32_t *vmask_ge; // [B][N][P]
  int32_t *nvalid;    // [B][N]
  int32_t *orig;      // [B][N][P]
  // association
  double *pair;       // [B][pair_stride][P][P]
  int32_t *assoc;     // [B][H][N]
  int32_t *nviews;    // [B][H]
  int32_t *nhyp;      // [B]
  double *C;          // [B][H*P]
  uint8_t *mask;      // [B][H*P]
  double *mwork;      // [B][H*P] Munkres working copy
  uint32_t *flags;    // [B]
  // triangulation + output
  se3d_person3d *hypP;   // [B][H]
  uint8_t *keep;         // [B][H]
  uint32_t *tri_list;    // [B * H * 17] (frame * H + hypothesis) << 5 | joint of every joint the fast triangulation kernels hand to the second pass
  uint32_t *tri_cnt;     // [8] ([4], [5]: camera steps of the per-camera sweep with hypotheses and detections / those that ran the Hungarian, running totals); [0..3]: entries of tri_list of the current launch; [1]: running total (statistics, se3d_get_exact_stats); [2]: joints triangulated
  float *tri_ws;         // [tri_waves][se3d_exact_ws_floats(N)] lane-interleaved rows x 4 workspaces of the float32 solves (8 N x 64 floats; + the centred samples on large rigs)
  se3d_person3d *outP;   // [B][H]
  int32_t *out_count;    // [B]
  int32_t *out_hyp;      // [B][H]
  int32_t *out_assoc;    // [B][H][N] original detection indices
};

__host__ __device__ inline int se3d_pair_index(int i, int j, int n) {  // get_fundamental_idx (S3D:242-253), closed form, i < j
  return i * n - i * (i + 1) / 2 + (j - i - 1);
}

// model tables (S3D:100-103, 130-132, 139-145); index [pose_method]
__device__ __constant__ const int c_kpParent[2][SE3D_NKP] = {{-1, 0, 0, 1, 2, 0, 0, 5, 6, 7, 8, 5, 6, 11, 12, 13, 14},
                                                            {-1, 0, 0, 2, 3, 2, 2, 5, 6, 7, 8, 4, 4, 11, 12, 13, 14}};
__device__ __constant__ const double c_limbLength[2][SE3D_NKP] = {
    {-1, 0.05, 0.05, 0.10, 0.10, -1, -1, 0.28, 0.28, 0.25, 0.25, 0.50, 0.50, 0.45, 0.45, 0.446, 0.446},
    {-1, 0.115, 0.116, 0.255, 0.238, 0.149, 0.149, 0.28, 0.28, 0.25, 0.25, 0.134, 0.134, 0.449, 0.449, 0.446, 0.446}};
__device__ __constant__ const double c_limbLSigma[2][SE3D_NKP] = {
    {-1, 0.05, 0.05, 0.05, 0.05, -1, -1, 0.10, 0.10, 0.10, 0.10, 0.15, 0.15, 0.10, 0.10, 0.10, 0.10},
    {-1, 0.07, 0.07, 0.15, 0.15, 0.10, 0.10, 0.15, 0.15, 0.15, 0.15, 0.10, 0.10, 0.20, 0.20, 0.20, 0.20}};
__device__ __constant__ const int c_kp2fusion[2][SE3D_NKP] = {{0, 16, 15, 18, 17, 5, 2, 6, 3, 7, 4, 12, 9, 13, 10, 14, 11},
                                                             {0, 19, 1, 20, 8, 5, 2, 6, 3, 7, 4, 12, 9, 13, 10, 14, 11}};

// FUSION_BODY_PARTS indices used by the post-processing (FBP:3-24)
enum { FU_Nose = 0, FU_Neck = 1, FU_RShoulder = 2, FU_LShoulder = 5, FU_MidHip = 8, FU_RHip = 9, FU_RAnkle = 11, FU_LHip = 12, FU_LAnkle = 14, FU_Belly = 20 };

// ---------------------------------------------------------------------------------------------------------------
// stage 2 (pose_prior) device state.  S = sequences (one pose_prior node instance each, = frame slots of stage 1),
// T = max tracks per sequence.  Track order (the reference's g_tracks vector order, which fixes Hungarian column order
// and merge precedence) is the indirection order[s][k] -> slot; erasing a track is a stable compaction of `order`.
// ---------------------------------------------------------------------------------------------------------------
struct Se3dStage2 {
  int S, T, H, N;
  int pose_method, norm_height;
  // inputs of the current call
  const se3d_person3d *dets;  // [S][H]
  const int32_t *n_det;       // [S]
  const float *fb_delay;      // [S][N]
  const double *t;            // [S]
  int seq0;                   // first sequence handled by this launch (blockIdx + seq0)
  // per-sequence globals (PP:54-60)
  double *fb_buf;     // [S][3]
  double *t_prev;     // [S]
  int32_t *next_id;   // [S]
  int32_t *frame_nr;  // [S]
  int32_t *n_tracks;  // [S]
  int32_t *order;     // [S][T]
  double *pred_dt;    // [S]
  // per-track slots (PP:68-76)
  uint8_t *used;        // [S][T]
  int32_t *id;          // [S][T]
  int32_t *num_obs;     // [S][T]
  uint32_t *has;        // [S][T] bit k: prevEstimate.exists(k)
  double *trk_t_prev;   // [S][T]
  double *height_prev;  // [S][T]
  double *root_prev;    // [S][T][3]
  double *prev;         // [S][T][21][3]
  double *vel;          // [S][T][21][3][3]
  // per-call work
  double *C;            // [S][H*T] cost matrix (rows = detections), column-major ld = n_det
  double *W;            // [S][H*T] Munkres working copy
  unsigned long long *near;  // [S][T][4] bit j of row i: tracks in slots i, j are closer than g_merge_dist_thresh (PP:872)
  int32_t *det_slot;    // [S][H] track slot of each detection (-1: none)
  uint8_t *pub;         // [S][H] publish flag of each detection
  se3d_person3d *fused_det;  // [S][H] per detection
  se3d_person3d *pred_det;   // [S][H]
  int32_t *iters_det;        // [S][H]
  double *obj_det;           // [S][H] final LM objective per detection (0.5 sum ||whitened r||^2 at the published values; -1: the optimisation failed)
  // outputs
  se3d_person3d *fused;   // [S][H] compacted
  se3d_person3d *pred;    // [S][H]
  int32_t *iters;         // [S][H]
  double *obj;            // [S][H] final LM objective of each published person (se3d_fused_out.lm_objective)
  int32_t *n_out;         // [S]
  uint32_t *flags2;       // [S]
  double *lm_scratch;     // [S][4][1865] HBM Schur workspace of the fused walk's non-forest LM fallback
  int shard_rank, shard_world;  // hypothesis shard: the per-step LM kernel refines detections i with i % world == rank only
  int u_valid, u_fslot;          // fused walk, one sequence per workgroup: the step's per-sequence scalars, loaded once after the association
  double u_pred_dt, u_t_prev;    // phase instead of by every LM wavefront pass at output time (pred_dt[s], frame_nr[s] % 3, t_prev[s])
  int direct_out;       // fused walk: the association phase numbers the detections that will be published (pub = 1 + output slot) and the LM
                        // writes a published person straight into fused / pred / iters; nothing is written for an unpublished one
};

// One refined detection as it crosses the hypothesis-shard all-gather after the LM step: the two published payloads and the state of
// the track the detection updated (what the owner's se3d_k_lm_refine wrote), so that every rank ends the step with identical state.
struct Se3dShardRec {
  se3d_person3d fused, pred;
  double prev[SE3D_NFU * 3], vel[SE3D_NFU * 9], root[3], t_prev, height;
  int32_t iters, num_obs;
  uint32_t has;
  uint8_t pub, pad[3];
  double obj;
};

// Multi-step stage-2 walk (se3d_run_stage2_walk): step k consumes the stage-1 output of frame slots slot0[k] .. slot0[k]+S-1 and
// writes its fused / predicted persons into the walk buffers at the same frame slots.
#define SE3D_WALK_MAX_STEPS 64
struct Se3dWalk {
  int n_steps;
  int lds_state;               // set by the launcher: per-track state staged in LDS for the duration of the walk (bit 1: previous poses, bit 2: bookkeeping)
  int book_offset;             // byte offset of the bookkeeping block in the dynamic LDS (lds_state & 4)
  int slot0[SE3D_WALK_MAX_STEPS];
  const double *t;             // [n_steps][t_stride] (device)
  int t_stride;
  const se3d_person3d *outP;   // stage-1 output [B][H]
  const int32_t *out_count;    // [B]
  const float *fb;             // [B][N]
  se3d_person3d *Wf, *Wp;      // [B][H] fused / predicted persons per frame slot
  int32_t *Wit;                // [B][H] LM iterations
  double *Wobj;                // [B][H] final LM objectives
  int32_t *Wn;                 // [B] published persons per frame slot
  double *Wdt;                 // [B] pred_delta_t per frame slot
};

// ---------------------------------------------------------------------------------------------------------------
// feedback reprojection (pose_reprojection node).  N cameras, T = capacity in persons.
// ---------------------------------------------------------------------------------------------------------------
struct Se3dReproj {
  int N, T, n, pose_method;
  int dense_form;             // host side: 1 = the three-kernel form even where the per-camera lists fit LDS (options.reproj_form; parity tests)
  int in_stride;              // batched launch (grid.z = result slot b): persons of slot b at in + b * in_stride, count n_dev[b]; every output
                              // array below holds one block per slot (dense / valid / slot / out / out_src: N * T entries, det_offset: N + 1)
  const int32_t *n_dev;       // optional device-side person count (resident chaining after stage 2); overrides n
  const double *cam;          // [N][16]: rows of the 3x4 base->camera transform (12), then fx, fy, cx, cy
  const int32_t *wh;          // [N][2] image width, height
  const se3d_person3d *in;    // [<=T]
  se3d_person2d *dense;       // [N][T] projected person per (camera, person)
  uint8_t *valid;             // [T][N] 1: person has >= 1 keypoint inside camera's image
  int32_t *slot;              // [T][N] output slot or -1
  int32_t *det_offset;        // [N+1]
  se3d_person2d *out;         // [N*T] compacted, camera-major, person order preserved (REPROJ:226-233)
  int32_t *out_src;           // [N*T] input person index of each output entry
};
hipError_t se3d_launch_reproject(const Se3dReproj &R, int n_grid, hipStream_t stream, int batch = 1);
size_t se3d_reproj_lists_lds(int N, int T, int dense_form);   // > 0: small rig, the one-kernel form runs and dense / valid / slot are not touched
