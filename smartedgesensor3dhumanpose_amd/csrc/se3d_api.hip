// se3d_api.hip — host side of libse3d_hip.so: context, HBM residency, staging and the C-ABI of include/se3d.h.
// The product path runs ONLY the HIP kernels; there is no CPU fallback (a missing/failed device is an error).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <chrono>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "se3d_device.h"

extern "C" void se3d_launch_munkres(double *d_dist, int *d_assign, int *d_status, int nR, int nC, hipStream_t st, int variant);
extern "C" void se3d_launch_pack_dense(const se3d_person3d *fused, const int32_t *n_out, int n_frames, int H, int capacity, int32_t *off, se3d_person3d *dense, hipStream_t st);
extern "C" void se3d_launch_stage1(const Se3dDev &D, const se3d_options &o, const int2 *pair_ab, hipStream_t st, hipEvent_t *ev, int phases, hipStream_t st_hi,
                                   hipEvent_t *ev_hi, const Se3dPairPlan *plan, hipEvent_t *ev_chunk);
extern "C" void se3d_launch_normalize_keep(const Se3dDev &D, hipStream_t st);
extern "C" void se3d_launch_pair_block(const Se3dDev &D, const int2 *d_ab_one, int frame, hipStream_t st);
extern "C" void se3d_launch_stage2(const Se3dStage2 &Q, const se3d_options &o, int seq0, int nseq, hipStream_t st, hipEvent_t *ev, int phases);
extern "C" void se3d_launch_shard_pack_hyp(const Se3dDev &D, se3d_person3d *out, hipStream_t st);
extern "C" void se3d_launch_shard_unpack_hyp(const Se3dDev &D, const se3d_person3d *gathered, hipStream_t st);
extern "C" void se3d_launch_shard_pack_lm(const Se3dStage2 &Q, int nseq, Se3dShardRec *out, hipStream_t st);
extern "C" void se3d_launch_shard_unpack_lm(const Se3dStage2 &Q, int nseq, const Se3dShardRec *gathered, hipStream_t st);
extern "C" bool se3d_launch_stage2_walk(const Se3dStage2 &Q, const se3d_options &o, const Se3dWalk &Wk, int nseq, hipStream_t st);
extern "C" void se3d_launch_dlt(const Se3dDev &D, const se3d_options &o, const int32_t *d_assoc_orig, int nh, hipStream_t st, hipEvent_t *ev);
extern "C" void se3d_launch_stream_scatter(const Se3dDev &D, const se3d_person2d *rag, const int32_t *off, hipStream_t st);
extern "C" void se3d_launch_stream_compact(const se3d_person3d *fused, const se3d_person3d *pred, const int32_t *n_out, int n_frames, int H, int32_t *out_off,
                                           int32_t *out_off_mirror, se3d_person3d *dense_fused, se3d_person3d *dense_pred, hipStream_t st);
extern "C" hipError_t se3d_stage1_init_device(void);
extern "C" hipError_t se3d_stage2_init_device(void);
extern "C" hipError_t se3d_reproj_init_device(void);

#define SE3D_VERSION "se3d-hip 0.6.0 (gfx950)"

namespace {
enum ProfSlot { PS_NORMALIZE = 0, PS_PAIRS, PS_SWEEP, PS_TRI, PS_POST, PS_ASSOC2, PS_LM, PS_UPDATE2, PS_WALK, PS_DLT, PS_TRI_EXACT, PS_COUNT };
const int kStage1Slots = 5, kStage2Slots = 3;
const char *kProfNames[PS_COUNT] = {"normalize", "epipolar_pairs", "sweep", "triangulate", "postprocess", "track_assoc", "lm_refine", "track_update", "stage2_walk", "dlt", "triangulate_exact"};
}  // namespace

struct se3d_ctx {
  Se3dDev D{};
  se3d_params params{};
  se3d_limits limits{};
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream_hi = nullptr;   // high-priority stream for the latency-bound chains of large problems (per-camera sweep, per-step stage 2)
  hipEvent_t ev_hi[4]{};
  Se3dPairPlan plan{};               // camera-column chunks of the pair table (large rigs; plan.m == 0: whole table)
  hipEvent_t ev_chunk[2 * SE3D_PAIR_CHUNKS_MAX]{};   // [j]: pair chunk j written, [MAX + j]: sweep of chunk j done (buffer j % 2 free again)
  hipEvent_t *cur_ev1 = nullptr;
  std::string err;
  std::vector<void *> allocs;
  int2 *pair_ab = nullptr;
  // Slabs: the arrays that cross the bus together are carved from ONE device allocation and ONE pinned host allocation at the same
  // offsets, so that the transfer of a whole batch (a single frame for max_frames = 1, the reference's operating point) is one copy
  // instead of one per array: in1 = {nraw, fb, raw}, out1 = {out_count, nhyp, flags, outP, out_hyp, out_assoc},
  // in2 = {n_det, t, fb_ref, dets}, out2 = {n_out, n_tracks, flags2, pred_dt, obj, iters, fused, pred}.
  struct Slab { char *d = nullptr, *h = nullptr; size_t bytes = 0; } sl_in1, sl_out1, sl_in2, sl_out2;
  size_t off_raw = 0, off_in2 = 0;   // offsets of the variable-length tails of the two input slabs
  // pinned host staging (views into the slabs)
  se3d_person2d *h_raw = nullptr;
  int32_t *h_nraw = nullptr;
  se3d_person3d *h_out = nullptr;
  int32_t *h_count = nullptr, *h_outhyp = nullptr, *h_outassoc = nullptr, *h_nhyp = nullptr;
  uint32_t *h_flags = nullptr;
  int staged_frames = 0;
  bool ran = false;
  se3d_options opt{};        // se3d_create_ex / se3d_set_options; all zero = defaults
  unsigned opt_version = 0;  // bumped by se3d_set_options: a captured call graph of an older version is captured again
  // profiling
  bool prof = false;
  static constexpr int kRing = 96;       // event sets in flight before the oldest are harvested
  hipEvent_t ev[kRing][7]{};             // stage-1 sets (event 6: between the first and the second pass of the triangulation)
  bool ev_dlt[kRing]{};                  // set brackets one config-2 pass (events 0 and 1 only)
  int32_t *d_assoc_in = nullptr;         // [B][H][N] caller's association table (config 2, se3d_stage_assoc)
  int dlt_hyp = 0;                       // hypotheses per frame of the staged table (0: none staged)
  bool ran_dlt = false;
  int ev_head = 0, ev_pending = 0;
  double prof_ms[PS_COUNT]{};
  int64_t prof_n[PS_COUNT]{};
  // stage 2
  Se3dStage2 Q{};
  se3d_person3d *d_in2 = nullptr;  // [S][H] host-provided measurements (se3d_refine)
  int32_t *d_ndet2 = nullptr;      // [S]
  float *d_fb = nullptr;           // [S][N] fb_delay of the staged frames (resident path)
  float *d_fb_ref = nullptr;       // [S][N] fb_delay rows of se3d_refine (host path): its own buffer, so mixing the two paths on one
                                   // context cannot overwrite the staged batch's values
  float *h_fb_ref = nullptr;
  se3d_person3d *h_in2 = nullptr;  // pinned mirrors of d_in2 / d_ndet2 / d_t (in2 slab)
  int32_t *h_ndet2 = nullptr;
  double *h_t_ref = nullptr;
  double *d_hung = nullptr;        // se3d_hungarian scratch: 256 x 256 cost matrix + assignment + status
  int *d_hung_assign = nullptr;
  double *d_t = nullptr;           // [S]
  float *h_fb = nullptr;
  double *h_t = nullptr;                 // [kRing][S] ring: a slot is reused only after its H2D copy has completed
  hipEvent_t t_done[kRing]{};
  int t_head = 0;
  se3d_person3d *h_fused = nullptr, *h_pred = nullptr;
  int32_t *h_iters = nullptr, *h_nout = nullptr, *h_ntracks = nullptr;
  double *h_obj = nullptr;
  uint32_t *h_flags2 = nullptr;
  double *h_preddt = nullptr;
  hipEvent_t ev2[kRing][4]{};            // stage-2 sets
  bool ev2_walk[kRing]{};                // set brackets one fused multi-step launch (events 0 and 3 only)
  int ev2_head = 0, ev2_pending = 0;
  bool ran2 = false;
  // streaming ingest (se3d_stream_submit / se3d_stream_collect): two batches in flight
  struct StreamSlot {
    se3d_person2d *h_rag = nullptr, *d_rag = nullptr;   // ragged input, pinned host / device
    int32_t *h_off = nullptr, *d_off = nullptr;         // [B][N + 1] absolute person offsets
    se3d_person3d *d_fused = nullptr, *d_pred = nullptr, *h_fused = nullptr, *h_pred = nullptr;  // dense published persons
    se3d_person3d *hd_fused = nullptr, *hd_pred = nullptr;   // device-side addresses of the pinned h_fused / h_pred (the compaction kernel writes them directly)
    int32_t *hd_out_off = nullptr;                           // ... and of h_out_off
    bool direct = false;                                     // this batch was written straight into the pinned buffers (options.stream_out = 1)
    int32_t *d_out_off = nullptr, *h_out_off = nullptr; // [B + 1]
    float *h_fb = nullptr, *d_fb = nullptr;             // [B][N] fb_delay of the batch (per slot: batch k + 1 is copied while batch k walks)
    hipEvent_t in_done = nullptr, compute_done = nullptr, off_done = nullptr, d2h_start = nullptr;
    int n_frames = 0;
    bool busy = false;
  } ss[2];
  bool dry = false;          // dev_alloc only counts (se3d_device_bytes)
  size_t dry_bytes = 0;
  bool stream_ready = false;
  int ss_head = 0, ss_tail = 0;  // next slot to submit into / oldest batch not yet collected
  // One frame per call (max_frames = 1, the reference's operating point): upload + kernels + download of se3d_triangulate / se3d_refine as
  // ONE hipGraph each, captured on first use and replayed (round 4).  Re-captured when the A/B switches the launchers read change.
  struct CallGraph { hipGraphExec_t exec = nullptr; unsigned version = 0; bool failed = false; } g1, g2;
  Se3dStage2 Q_shard{};   // views of the se3d_run_stage2_shard_pre call, reused by its _post half
  const float *d_fb_walk = nullptr;  // fb_delay rows the next se3d_run_stage2_walk reads (streaming: the slot's; else d_fb)
  // feedback reprojection
  Se3dReproj R{};
  se3d_person3d *d_rp_in = nullptr;
  int32_t *h_rp_off = nullptr;
  bool rp_ready = false;
  Se3dReproj RB{};          // batch buffers of se3d_reproject_resident_batch (one block of every output array per result slot)
  int rpb_cap = 0, rpb_slot0 = 0, rpb_n = 0;
  bool rpb_lists = false;   // the batch buffers were sized for the one-kernel form (no dense intermediate)
};

namespace {

bool hip_ok(se3d_ctx *c, hipError_t e, const char *what) {
  if (e == hipSuccess) return true;
  c->err = std::string(what) + ": " + hipGetErrorString(e);
  return false;
}
#define HIPCHK(c, call)                                  \
  do {                                                   \
    if (!hip_ok((c), (call), #call)) return SE3D_ERR_DEVICE; \
  } while (0)

template <typename T>
bool dev_alloc(se3d_ctx *c, T **p, size_t n) {
  if (c->dry) {   // se3d_device_bytes: the allocation list of se3d_create is walked without touching a device
    c->dry_bytes += (n * sizeof(T) + 255) & ~(size_t)255;
    *p = nullptr;
    return true;
  }
  void *q = nullptr;
  if (!hip_ok(c, hipMalloc(&q, n * sizeof(T)), "hipMalloc")) return false;
  c->allocs.push_back(q);
  *p = reinterpret_cast<T *>(q);
  return true;
}

// Slab construction: `carve` names the arrays once; it runs with null bases to size the slab and again with the real bases.
struct SlabCarver {
  char *d, *h;
  size_t off = 0;
  template <typename T>
  size_t take(T **dp, T **hp, size_t n) {
    const size_t at = off;
    if (dp) *dp = d ? reinterpret_cast<T *>(d + off) : nullptr;
    if (hp) *hp = h ? reinterpret_cast<T *>(h + off) : nullptr;
    off += (n * sizeof(T) + 255) & ~(size_t)255;
    return at;
  }
};
template <typename F>
bool slab_build(se3d_ctx *c, se3d_ctx::Slab &s, F carve) {
  SlabCarver sizing{nullptr, nullptr};
  carve(sizing);
  s.bytes = sizing.off;
  if (!dev_alloc(c, &s.d, s.bytes)) return false;
  if (c->dry) return true;
  if (!hip_ok(c, hipHostMalloc((void **)&s.h, s.bytes, hipHostMallocDefault), "hipHostMalloc (slab)")) return false;
  memset(s.h, 0, s.bytes);
  SlabCarver real{s.d, s.h};
  carve(real);
  return true;
}

// harvest completed event sets (all of them if `all`, else only enough to free one slot of the ring)
void collect_profile(se3d_ctx *c, bool all = true) {
  while (c->ev_pending > 0 && (all || c->ev_pending >= se3d_ctx::kRing)) {
    const int idx = (c->ev_head - c->ev_pending + 2 * se3d_ctx::kRing) % se3d_ctx::kRing;
    if (c->ev_dlt[idx]) {
      float ms = 0.f;
      (void)hipEventSynchronize(c->ev[idx][1]);
      if (hipEventElapsedTime(&ms, c->ev[idx][0], c->ev[idx][1]) == hipSuccess) { c->prof_ms[PS_DLT] += ms; c->prof_n[PS_DLT] += 1; }
      --c->ev_pending;
      continue;
    }
    (void)hipEventSynchronize(c->ev[idx][5]);
    for (int i = 0; i < kStage1Slots; ++i) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, c->ev[idx][i], c->ev[idx][i + 1]) == hipSuccess) { c->prof_ms[i] += ms; c->prof_n[i] += 1; }
    }
    {   // the float32 second pass of the triangulation on its own (its time is also inside "triangulate")
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, c->ev[idx][6], c->ev[idx][4]) == hipSuccess) { c->prof_ms[PS_TRI_EXACT] += ms; c->prof_n[PS_TRI_EXACT] += 1; }
    }
    --c->ev_pending;
  }
  while (c->ev2_pending > 0 && (all || c->ev2_pending >= se3d_ctx::kRing)) {
    const int idx = (c->ev2_head - c->ev2_pending + 2 * se3d_ctx::kRing) % se3d_ctx::kRing;
    (void)hipEventSynchronize(c->ev2[idx][3]);
    if (c->ev2_walk[idx]) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, c->ev2[idx][0], c->ev2[idx][3]) == hipSuccess) { c->prof_ms[PS_WALK] += ms; c->prof_n[PS_WALK] += 1; }
    } else
    for (int i = 0; i < kStage2Slots; ++i) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, c->ev2[idx][i], c->ev2[idx][i + 1]) == hipSuccess) { c->prof_ms[kStage1Slots + i] += ms; c->prof_n[kStage1Slots + i] += 1; }
    }
    --c->ev2_pending;
  }
}
hipEvent_t *next_ev1(se3d_ctx *c) {
  if (!c->prof) return nullptr;
  collect_profile(c, false);
  hipEvent_t *e = c->ev[c->ev_head];
  c->ev_dlt[c->ev_head] = false;
  c->ev_head = (c->ev_head + 1) % se3d_ctx::kRing;
  ++c->ev_pending;
  return e;
}
hipEvent_t *next_ev2(se3d_ctx *c) {
  if (!c->prof) return nullptr;
  collect_profile(c, false);
  hipEvent_t *e = c->ev2[c->ev2_head];
  c->ev2_walk[c->ev2_head] = false;
  c->ev2_head = (c->ev2_head + 1) % se3d_ctx::kRing;
  ++c->ev2_pending;
  return e;
}

int stage2_reset(se3d_ctx *c) {
  const Se3dStage2 &Q = c->Q;
  const size_t S = Q.S, T = Q.T;
  if (hipMemsetAsync(Q.n_tracks, 0, sizeof(int32_t) * S, c->stream) != hipSuccess) return SE3D_ERR_DEVICE;
  (void)hipMemsetAsync(Q.used, 0, S * T, c->stream);
  (void)hipMemsetAsync(Q.next_id, 0, sizeof(int32_t) * S, c->stream);
  (void)hipMemsetAsync(Q.frame_nr, 0, sizeof(int32_t) * S, c->stream);
  (void)hipMemsetAsync(Q.t_prev, 0, sizeof(double) * S, c->stream);
  (void)hipMemsetAsync(Q.flags2, 0, sizeof(uint32_t) * S, c->stream);
  (void)hipMemsetAsync(Q.n_out, 0, sizeof(int32_t) * S, c->stream);
  std::vector<double> fb(S * 3, 0.10);  // g_fb_delay_buffer(g_n_mov_avg, g_avg_delay) (PP:54)
  if (hipMemcpyAsync(Q.fb_buf, fb.data(), sizeof(double) * S * 3, hipMemcpyHostToDevice, c->stream) != hipSuccess) return SE3D_ERR_DEVICE;
  if (hipStreamSynchronize(c->stream) != hipSuccess) return SE3D_ERR_DEVICE;
  c->ran2 = false;
  return SE3D_OK;
}

// 3x4 pseudo-inverse the way the reference forms it (pseudo_inv34d, S3D:236-240): P = U S V^T by a Jacobi SVD, singular values below
// eps * max(rows, cols) * s_max dropped, P^+ = V S^+ U^T.  Host, fp64.  The SVD is a one-sided (Hestenes) Jacobi iteration on the 4x3
// matrix B = P^T: column pairs are rotated until they are mutually orthogonal (B -> U' S, rotations accumulated in V'), so that
// P^T = U' S V'^T and P^+ = U' S^+ V'^T.  Every operation is spelled out in a fixed order and the file is compiled with
// -ffp-contract=off: the test oracle forms its matrices by the same sequence and the two F tables are compared bit for bit
// (tests/test_abi.py) — F feeds a strict `<` against max_epi_dist (S3D:620), so "close" is not good enough for index parity.
void pinv34(const double P[12], double out[12]) {
  double B[4][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int r = 0; r < 4; ++r)
    for (int k = 0; k < 3; ++k) B[r][k] = P[k * 4 + r];
  for (int sweep = 0; sweep < 60; ++sweep) {
    double worst = 0.0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double pp = 0, qq = 0, pq = 0;
        for (int r = 0; r < 4; ++r) {
          pp += B[r][p] * B[r][p];
          qq += B[r][q] * B[r][q];
          pq += B[r][p] * B[r][q];
        }
        if (pq == 0.0) continue;
        worst = std::max(worst, std::fabs(pq) / std::sqrt(pp * qq));
        const double zeta = (qq - pp) / (2.0 * pq);
        const double tn = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        const double cs = 1.0 / std::sqrt(1.0 + tn * tn), sn = cs * tn;
        for (int r = 0; r < 4; ++r) {
          const double x = B[r][p], y = B[r][q];
          B[r][p] = cs * x - sn * y;
          B[r][q] = sn * x + cs * y;
        }
        for (int r = 0; r < 3; ++r) {
          const double x = V[r][p], y = V[r][q];
          V[r][p] = cs * x - sn * y;
          V[r][q] = sn * x + cs * y;
        }
      }
    if (worst < 1e-16) break;
  }
  double sv[3];
  for (int k = 0; k < 3; ++k) {
    double nn = 0;
    for (int r = 0; r < 4; ++r) nn += B[r][k] * B[r][k];
    sv[k] = std::sqrt(nn);
    if (sv[k] > 0)
      for (int r = 0; r < 4; ++r) B[r][k] /= sv[k];
  }
  const double tol = std::numeric_limits<double>::epsilon() * 4.0 * std::max(sv[0], std::max(sv[1], sv[2]));
  for (int r = 0; r < 4; ++r)
    for (int j = 0; j < 3; ++j) {
      double acc = 0;
      for (int k = 0; k < 3; ++k) acc += B[r][k] * ((std::fabs(sv[k]) > tol) ? 1.0 / sv[k] : 0.0) * V[j][k];
      out[r * 3 + j] = acc;
    }
}

// Why the last se3d_create of this thread returned NULL: kept for se3d_last_error(NULL) (a NULL context has no error slot of its own)
// and printed to stderr as before.
static thread_local std::string t_create_err;
static void create_error(const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  t_create_err = buf;
  fprintf(stderr, "%s\n", buf);
}

// Every device buffer a context owns from its creation on (the streaming slots, the reprojection lists and the hypothesis-shard blocks are
// allocated by the first call that needs them).  One list for se3d_create and for se3d_device_bytes (dry run: c->dry).
struct RigDev { float *Rt = nullptr, *K = nullptr, *F = nullptr; se3d_person2d *raw = nullptr; int32_t *nraw = nullptr; };
static bool alloc_device_state(se3d_ctx *c, int N, const se3d_params *params, const se3d_limits *limits, RigDev &r) {
  const int B = limits->max_frames, P = limits->max_det_per_cam, H = limits->max_hyp;
  Se3dDev &D = c->D;
  const int np = N * (N - 1) / 2;
  D.d = Se3dDims{B, N, P, H, np, params->pose_method, params->norm_height, params->max_epi_dist, 0, 1};
  bool ok = true;
  ok = ok && dev_alloc(c, &r.Rt, (size_t)N * 12) && dev_alloc(c, &r.K, (size_t)N * 4) && dev_alloc(c, &r.F, (size_t)np * 9);
  ok = ok && slab_build(c, c->sl_in1, [&](SlabCarver &m) {   // raw last: one copy covers the header and the staged frames
    m.take(&r.nraw, &c->h_nraw, (size_t)B * N);
    m.take(&c->d_fb, &c->h_fb, (size_t)B * N);
    c->off_raw = m.take(&r.raw, &c->h_raw, (size_t)B * N * P);
  });
  ok = ok && dev_alloc(c, &D.kp, (size_t)B * N * P * SE3D_NKP) && dev_alloc(c, &c->d_assoc_in, (size_t)B * H * N);
  ok = ok && dev_alloc(c, &D.pscore, (size_t)B * N * P) && dev_alloc(c, &D.vmask_gt, (size_t)B * N * P) && dev_alloc(c, &D.vmask_ge, (size_t)B * N * P);
  ok = ok && dev_alloc(c, &D.nvalid, (size_t)B * N) && dev_alloc(c, &D.orig, (size_t)B * N * P);
  {
    c->plan = se3d_pair_plan(N, se3d_pair_chunks_of(c->opt, N, P, H));
    D.d.pair_chunks = c->plan.m;
    D.d.pair_stride = c->plan.m ? 2 * c->plan.cap : np;
  }
  ok = ok && dev_alloc(c, &D.pair, (size_t)B * D.d.pair_stride * P * P);
  ok = ok && dev_alloc(c, &D.assoc, (size_t)B * H * N) && dev_alloc(c, &D.nviews, (size_t)B * H);
  ok = ok && dev_alloc(c, &D.C, (size_t)B * H * P) && dev_alloc(c, &D.mask, (size_t)B * H * P) && dev_alloc(c, &D.mwork, (size_t)B * H * P);
  {  // second pass of the triangulation (float32-exact solves of the flagged joints, se3d_k_triangulate_exact): list, counter, workspaces
    const size_t ws_floats = (size_t)64 * 8 * N + (N <= 64 ? 0 : (((size_t)3 * (4 * N + 1) + 63) & ~(size_t)63));   // se3d_exact_ws_floats (se3d_stage1.hip)
    const size_t per_wave = ws_floats * sizeof(float), joints = (size_t)B * H * SE3D_NKP;
    // one wavefront per 4 joints of the batch (the list is short: 0.1 % of the joints on clean scenes), 16 .. 4096 (16 per CU); <= 128 MB on
    // small rigs, <= 2 GB on large ones (a wavefront's workspace is 2 KB x cameras; the sigma samples of a joint seen by n views are 4 n + 1
    // solves of a 2 n x 4 system each — what the reference pays per joint too — so large rigs need the wavefronts: 2560 are resident at 256 cameras)
    D.d.tri_waves = (int)std::max<size_t>(1, std::min<size_t>(std::min<size_t>(4096, std::max<size_t>(16, joints / 4)), ((size_t)(N > 64 ? 2048 : 128) << 20) / per_wave));
    ok = ok && dev_alloc(c, &D.tri_list, joints) && dev_alloc(c, &D.tri_cnt, (size_t)8) && dev_alloc(c, &D.tri_ws, (size_t)D.d.tri_waves * ws_floats);
    if (ok && !c->dry) ok = hip_ok(c, hipMemset(D.tri_cnt, 0, 8 * sizeof(uint32_t)), "hipMemset (tri_cnt)");
  }
  ok = ok && dev_alloc(c, &D.hypP, (size_t)B * H) && dev_alloc(c, &D.keep, (size_t)B * H);
  ok = ok && slab_build(c, c->sl_out1, [&](SlabCarver &m) {
    m.take(&D.out_count, &c->h_count, (size_t)B);
    m.take(&D.nhyp, &c->h_nhyp, (size_t)B);
    m.take(&D.flags, &c->h_flags, (size_t)B);
    m.take(&D.outP, &c->h_out, (size_t)B * H);
    m.take(&D.out_hyp, &c->h_outhyp, (size_t)B * H);
    m.take(&D.out_assoc, &c->h_outassoc, (size_t)B * H * N);
  });
  ok = ok && dev_alloc(c, &c->pair_ab, (size_t)np);
  {  // stage 2 state: S = B sequences, T = max_tracks
    const int T = limits->max_tracks > 0 ? limits->max_tracks : H;
    Se3dStage2 &Q = c->Q;
    const size_t S = B;
    Q.S = B; Q.T = T; Q.H = H; Q.N = N; Q.pose_method = params->pose_method; Q.norm_height = params->norm_height; Q.seq0 = 0;
    Q.shard_rank = 0; Q.shard_world = 1;
    ok = ok && dev_alloc(c, &Q.fb_buf, S * 3) && dev_alloc(c, &Q.t_prev, S) && dev_alloc(c, &Q.next_id, S) && dev_alloc(c, &Q.frame_nr, S);
    ok = ok && dev_alloc(c, &Q.order, S * T);
    ok = ok && dev_alloc(c, &Q.used, S * T) && dev_alloc(c, &Q.id, S * T) && dev_alloc(c, &Q.num_obs, S * T) && dev_alloc(c, &Q.has, S * T);
    ok = ok && dev_alloc(c, &Q.trk_t_prev, S * T) && dev_alloc(c, &Q.height_prev, S * T) && dev_alloc(c, &Q.root_prev, S * T * 3);
    ok = ok && dev_alloc(c, &Q.prev, S * T * SE3D_NFU * 3) && dev_alloc(c, &Q.vel, S * T * SE3D_NFU * 9);
    ok = ok && dev_alloc(c, &Q.near, S * T * 4);
    ok = ok && dev_alloc(c, &Q.C, S * H * T) && dev_alloc(c, &Q.W, S * H * T) && dev_alloc(c, &Q.det_slot, S * H) && dev_alloc(c, &Q.pub, S * H);
    ok = ok && dev_alloc(c, &Q.fused_det, S * H) && dev_alloc(c, &Q.pred_det, S * H) && dev_alloc(c, &Q.iters_det, S * H) && dev_alloc(c, &Q.obj_det, S * H);
    ok = ok && dev_alloc(c, &Q.lm_scratch, S * 4 * 1865);  // [S][4 wavefronts][S2_SCHUR_SCRATCH_DOUBLES] (se3d_stage2.hip)
    ok = ok && slab_build(c, c->sl_out2, [&](SlabCarver &m) {
      m.take(&Q.n_out, &c->h_nout, S);
      m.take(&Q.n_tracks, &c->h_ntracks, S);
      m.take(&Q.flags2, &c->h_flags2, S);
      m.take(&Q.pred_dt, &c->h_preddt, S);
      m.take(&Q.obj, &c->h_obj, S * H);
      m.take(&Q.iters, &c->h_iters, S * H);
      m.take(&Q.fused, &c->h_fused, S * H);
      m.take(&Q.pred, &c->h_pred, S * H);
    });
    ok = ok && slab_build(c, c->sl_in2, [&](SlabCarver &m) {   // measurements last: one copy covers the header and the n_det persons (S = 1)
      m.take(&c->d_ndet2, &c->h_ndet2, S);
      m.take(&c->d_t, &c->h_t_ref, S);
      m.take(&c->d_fb_ref, &c->h_fb_ref, S * N);
      c->off_in2 = m.take(&c->d_in2, &c->h_in2, S * H);
    });
    ok = ok && dev_alloc(c, &c->d_hung, (size_t)SE3D_MUNKRES_MAX * SE3D_MUNKRES_MAX) &&
         dev_alloc(c, &c->d_hung_assign, (size_t)SE3D_MUNKRES_MAX + 1);
  }
  return ok;
}

// ---- hipGraph replay of the one-frame calls ------------------------------------------------------------------------------------------
// A max_frames = 1 context moves fixed-size slabs (in1 / out1, in2 / out2) and launches a fixed kernel chain with constant arguments: the whole
// call is captured once from the context's stream and replayed with one hipGraphLaunch — one submission instead of a copy, five to seven
// launches and a copy.  Not used while kernel timing is on (the HIP events of se3d_profile_* sit between the kernels), on rigs whose sweep runs
// on a second stream, or with options.graph = -1.  A capture is tied to the options version it was made under (se3d_set_options bumps it).
static bool graph_wanted(const se3d_ctx *c) {
  return !(c->prof || c->D.d.B != 1 || c->plan.m != 0 || c->D.d.shard_world > 1 || c->opt.graph < 0);
}
// Runs `enqueue` (which must only enqueue work on c->stream) through the graph `g`: captured on first use / after a switch change.
// Returns 1 if the graph was launched, 0 if the caller must enqueue directly (capture not possible), < 0 on a device error.
template <typename F>
static int graph_replay(se3d_ctx *c, se3d_ctx::CallGraph &g, F enqueue) {
  if (g.failed) return 0;
  if (g.exec && g.version != c->opt_version) { (void)hipGraphExecDestroy(g.exec); g.exec = nullptr; }   // options changed since the capture
  if (!g.exec) {
    hipGraph_t graph = nullptr;
    (void)hipGetLastError();
    if (hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { g.failed = true; (void)hipGetLastError(); return 0; }
    enqueue();
    // A launch the runtime rejects during capture (configuration error) leaves a graph WITHOUT that node: the direct path, which checks every
    // launch, must report it instead of a replay that silently skips a kernel.
    const hipError_t e0 = hipGetLastError();
    const hipError_t e1 = hipStreamEndCapture(c->stream, &graph);
    if (e0 != hipSuccess || e1 != hipSuccess || !graph || hipGraphInstantiate(&g.exec, graph, nullptr, nullptr, 0) != hipSuccess) {
      if (graph) (void)hipGraphDestroy(graph);
      g.exec = nullptr; g.failed = true; (void)hipGetLastError();
      return 0;
    }
    (void)hipGraphDestroy(graph);
    g.version = c->opt_version;
  }
  if (!hip_ok(c, hipGraphLaunch(g.exec, c->stream), "hipGraphLaunch")) return SE3D_ERR_DEVICE;
  return 1;
}

}  // namespace

extern "C" {

const char *se3d_version(void) { return SE3D_VERSION; }
int32_t se3d_abi_version(void) { return SE3D_ABI_VERSION; }

// Host set-up of the static rig tensors (replaces S3D:1187-1211).  F_ij = [P_j C_i]_x P_j P_i^+ in fp64, stored as float.
int se3d_rig_from_transforms(int32_t n_cam, const double *T, float *Rt_out, float *F_out) {
  if (n_cam < 2 || !T) return SE3D_ERR_ARG;
  if (Rt_out)
    for (int i = 0; i < n_cam; ++i)
      for (int k = 0; k < 12; ++k) Rt_out[12 * (size_t)i + k] = (float)T[16 * (size_t)i + k];
  if (F_out) {
    size_t idx = 0;
    for (int i = 0; i < n_cam; ++i) {
      const double *Mi = T + 16 * (size_t)i;
      double Ci[4];
      for (int a = 0; a < 3; ++a) Ci[a] = -(Mi[0 * 4 + a] * Mi[3] + Mi[1 * 4 + a] * Mi[7] + Mi[2 * 4 + a] * Mi[11]);
      Ci[3] = 1.0;
      double Pinv[12];
      pinv34(Mi, Pinv);
      for (int j = i + 1; j < n_cam; ++j, ++idx) {
        const double *Pj = T + 16 * (size_t)j;
        double e[3];
        for (int a = 0; a < 3; ++a) {
          e[a] = 0;
          for (int k = 0; k < 4; ++k) e[a] += Pj[a * 4 + k] * Ci[k];
        }
        const double ex[9] = {0, -e[2], e[1], e[2], 0, -e[0], -e[1], e[0], 0};
        // `e_ij_cross * Ps[j] * Pinv` (S3D:1201) associates to the left: (e_x P_j) P_i^+
        double M[12];
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 4; ++b) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += ex[a * 3 + k] * Pj[k * 4 + b];
            M[a * 4 + b] = s;
          }
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += M[a * 4 + k] * Pinv[k * 3 + b];
            F_out[9 * idx + a * 3 + b] = (float)s;
          }
      }
    }
  }
  return SE3D_OK;
}

static bool options_valid(const se3d_options *o) {
  if (!o) return true;
  for (int32_t r : o->reserved) if (r != 0) return false;
  return o->pair_chunks >= -1 && o->pair_chunks != 1 && o->pair_chunks <= SE3D_PAIR_CHUNKS_MAX && o->hi_priority_stream >= -1 && o->hi_priority_stream <= 0 && o->reserve_cus >= -1 &&
         o->graph >= -1 && o->graph <= 0 && o->walk_shape >= 0 && o->walk_shape <= SE3D_WALK_WG && o->sweep_lds_kb >= 0 && o->sweep_lds_kb <= 128 &&
         o->pairs_form >= 0 && o->pairs_form <= SE3D_PAIRS_FUSED && o->sweep_form >= 0 && o->sweep_form <= SE3D_SWEEP_WAVE_GLOBAL && o->tri_form >= 0 && o->tri_form <= SE3D_TRI_JOINT &&
         o->post_form >= 0 && o->post_form <= SE3D_POST_BLOCK && o->lm_form >= 0 && o->lm_form <= SE3D_LM_DENSE && o->walk_form >= 0 && o->walk_form <= SE3D_WALK_STEPS &&
         o->exact_cov >= 0 && o->exact_cov <= SE3D_EXACT_COV_F32 && o->walk_lm >= 0 && o->walk_lm <= SE3D_WALK_LM_SOLO && o->walk_book >= 0 && o->walk_book <= 1 && o->refine_steps >= 0 && o->refine_steps <= 1 &&
         o->munkres_form >= 0 && o->munkres_form <= SE3D_MUNKRES_WAVE && o->reproj_form >= 0 && o->reproj_form <= 1 && o->stream_out >= 0 && o->stream_out <= 1;
}

se3d_ctx *se3d_create(const se3d_rig *rig, const se3d_params *params, const se3d_limits *limits, int32_t device_id) {
  return se3d_create_ex(rig, params, limits, nullptr, device_id);
}

se3d_ctx *se3d_create_ex(const se3d_rig *rig, const se3d_params *params, const se3d_limits *limits, const se3d_options *options, int32_t device_id) {
  if (!rig || !params || !limits || !rig->Rt || !rig->K || !rig->F) return nullptr;
  if (!options_valid(options)) { create_error("se3d_create_ex: an option is out of range (include/se3d.h: se3d_options)"); return nullptr; }
  const int N = rig->n_cam, B = limits->max_frames, P = limits->max_det_per_cam, H = limits->max_hyp;
  if (N < 2 || N > SE3D_MAX_CAMS || B < 1 || P < 1 || P > SE3D_MUNKRES_MAX || H < P || H > SE3D_MUNKRES_MAX) {
    create_error("se3d_create: unsupported sizes (n_cam %d [2..%d], det/cam %d [1..%d], hyp %d [det..%d])", N, SE3D_MAX_CAMS, P,
                 SE3D_MUNKRES_MAX, H, SE3D_MUNKRES_MAX);
    return nullptr;
  }
  if ((int64_t)B * H >= ((int64_t)1 << 25)) {   // tri_list entries: (frame * max_hyp + hypothesis) << 5 | joint, bit 31 = the covariance flag, bit 30 = internal mark
    create_error("se3d_create: max_frames x max_hyp = %lld exceeds 2^25 - 1 hypothesis slots per context", (long long)B * H);
    return nullptr;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device_id < 0 || device_id >= ndev) {
    create_error("se3d_create: no usable HIP device (count=%d, requested=%d); there is no CPU fallback", ndev, device_id);
    return nullptr;
  }
  se3d_ctx *c = new se3d_ctx();
  c->params = *params;
  c->limits = *limits;
  if (options) c->opt = *options;
  c->device = device_id;
  if (!hip_ok(c, hipSetDevice(device_id), "hipSetDevice")) { create_error("se3d_create: %s", c->err.c_str()); delete c; return nullptr; }
  {
    // Large rigs (>= 128 cameras) with the WHOLE pair table (options.pair_chunks = -1; the default since round 3 is column chunks without a
    // reservation): the bulk stream of the context (pair-cost kernel: millions of workgroups, 39 KB of LDS each) is kept off 32 CUs — the
    // mask's bits are spread round-robin over the eight XCDs, so that is ~4 CUs of every XCD, not one XCD — so that the small latency-bound
    // launches of the camera sweep and the per-step stage 2 (high-priority stream, all CUs; a sweep workgroup wants ~100 KB of LDS for its
    // Munkres matrix) always find a CU with room.  Measured at 256 cam x 128 p, four contexts: 21.0 k -> 22.5 k skeletons/s; 24 reserved CUs
    // lose (20.6 k).  options.reserve_cus = R overrides (-1 = none).
    // (column chunks of the pair table, round 3: pair kernel and sweep alternate in short launches and the Munkres matrix sits in L2 — no CUs reserved)
    const bool chunks = se3d_pair_chunks_of(c->opt, N, P, H) > 0;
    const int reserve = c->opt.reserve_cus > 0 ? (limits->max_hyp > 64 ? c->opt.reserve_cus : 0) : c->opt.reserve_cus < 0 ? 0 : (N >= 128 && !chunks ? 32 : 0);
    hipDeviceProp_t prop;
    bool ok_stream = false;
    if (reserve > 0 && hipGetDeviceProperties(&prop, device_id) == hipSuccess && reserve < prop.multiProcessorCount) {
      const int ncu = prop.multiProcessorCount;
      std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
      for (int i = 0; i < ncu - reserve; ++i) mask[i / 32] |= 1u << (i % 32);
      ok_stream = hipExtStreamCreateWithCUMask(&c->stream, (uint32_t)mask.size(), mask.data()) == hipSuccess;
    }
    if (!ok_stream && !hip_ok(c, hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking), "hipStreamCreate")) { delete c; return nullptr; }
  }
  if (limits->max_hyp > 64 && c->opt.hi_priority_stream >= 0) {  // large association problems: their launch chains get a high-priority queue
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    if (hipStreamCreateWithPriority(&c->stream_hi, hipStreamNonBlocking, hi) != hipSuccess) c->stream_hi = nullptr;
    for (auto &e : c->ev_hi) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
  }
  // dynamic-LDS limits of the large-LDS kernels: a per-device attribute, so it is set for every context's device
  if (!hip_ok(c, se3d_stage1_init_device(), "hipFuncSetAttribute (stage 1)") || !hip_ok(c, se3d_stage2_init_device(), "hipFuncSetAttribute (stage 2)") ||
      !hip_ok(c, se3d_reproj_init_device(), "hipFuncSetAttribute (reprojection)")) {
    create_error("se3d_create: %s", c->err.c_str());
    (void)hipStreamDestroy(c->stream);
    if (c->stream_hi) (void)hipStreamDestroy(c->stream_hi);
    for (auto &e : c->ev_hi)
      if (e) (void)hipEventDestroy(e);
    delete c;
    return nullptr;
  }
  for (auto &set : c->ev)
    for (auto &e : set) (void)hipEventCreate(&e);
  const int T_chk = limits->max_tracks > 0 ? limits->max_tracks : H;
  if (T_chk > SE3D_MUNKRES_MAX) { create_error("se3d_create: max_tracks %d > %d", T_chk, SE3D_MUNKRES_MAX); se3d_destroy(c); return nullptr; }
  Se3dDev &D = c->D;
  const int np = N * (N - 1) / 2;
  RigDev rd;
  bool ok = alloc_device_state(c, N, params, limits, rd);
  float *dRt = rd.Rt, *dK = rd.K, *dF = rd.F;
  se3d_person2d *draw = rd.raw;
  int32_t *dnraw = rd.nraw;
  {
    const size_t S = B;
    for (auto &set : c->ev2)
      for (auto &e : set) (void)hipEventCreate(&e);
    if (ok) {
      ok = ok && hip_ok(c, hipHostMalloc((void **)&c->h_t, sizeof(double) * S * se3d_ctx::kRing, hipHostMallocDefault), "hipHostMalloc");
      for (auto &e : c->t_done) (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
      ok = ok && hip_ok(c, hipMemset(c->d_fb, 0, sizeof(float) * S * N), "memset fb");
      ok = ok && (stage2_reset(c) == SE3D_OK);
    }
  }
  if (ok) {
    D.Rt = dRt; D.K = dK; D.F = dF; D.raw = draw; D.nraw = dnraw;
    std::vector<int2> ab(np);
    if (c->plan.m) {   // column-major: the pairs of a camera-column chunk are contiguous
      for (int j = 1, idx = 0; j < N; ++j)
        for (int i = 0; i < j; ++i, ++idx) ab[idx] = make_int2(i, j);
    } else {
      for (int i = 0, idx = 0; i < N; ++i)
        for (int j = i + 1; j < N; ++j, ++idx) ab[idx] = make_int2(i, j);
    }
    if (c->plan.m)
      for (auto &e : c->ev_chunk) ok = ok && hip_ok(c, hipEventCreateWithFlags(&e, hipEventDisableTiming), "hipEventCreate");
    ok = ok && hip_ok(c, hipMemcpy(dRt, rig->Rt, sizeof(float) * N * 12, hipMemcpyHostToDevice), "copy Rt");
    ok = ok && hip_ok(c, hipMemcpy(dK, rig->K, sizeof(float) * N * 4, hipMemcpyHostToDevice), "copy K");
    ok = ok && hip_ok(c, hipMemcpy(dF, rig->F, sizeof(float) * np * 9, hipMemcpyHostToDevice), "copy F");
    ok = ok && hip_ok(c, hipMemcpy(c->pair_ab, ab.data(), sizeof(int2) * np, hipMemcpyHostToDevice), "copy pair_ab");
    ok = ok && hip_ok(c, hipMemset(draw, 0, sizeof(se3d_person2d) * (size_t)B * N * P), "memset raw");
    ok = ok && hip_ok(c, hipMemset(dnraw, 0, sizeof(int32_t) * (size_t)B * N), "memset nraw");
  }
  if (!ok) {
    {
      size_t free_b = 0, total_b = 0;
      (void)hipMemGetInfo(&free_b, &total_b);
      create_error("se3d_create failed: %s (the context needs %.2f GB of device memory, %.2f of %.2f GB are free)", c->err.c_str(),
                   (double)se3d_device_bytes_ex(N, limits, &c->opt) / 1e9, free_b / 1e9, total_b / 1e9);
    }
    se3d_destroy(c);
    return nullptr;
  }
  // The fills above are hipMemset calls on the NULL stream: they return before the fill has run, and the context's streams are non-blocking, i.e.
  // they do not wait for the null stream.  Without this wait the zero fill of nraw could overtake the upload of the first se3d_stage_frames: the
  // pass then saw a frame set without detections (FEW_CAMERAS, no persons).  Latent since round 1, seen in round 5 in ~2 % of the runs once 14
  // processes shared the GPU (tests/test_gpu_fuzz_parity.py starts its runs side by side) and the fill was queued for milliseconds.
  if (!hip_ok(c, hipStreamSynchronize(nullptr), "hipStreamSynchronize (create-time fills)") || !hip_ok(c, hipStreamSynchronize(c->stream), "hipStreamSynchronize")) {
    create_error("se3d_create: %s", c->err.c_str());
    se3d_destroy(c);
    return nullptr;
  }
  return c;
}

int se3d_set_options(se3d_ctx *c, const se3d_options *o) {
  if (!c || !o || !options_valid(o)) return SE3D_ERR_ARG;
  if (o->pair_chunks != c->opt.pair_chunks || o->hi_priority_stream != c->opt.hi_priority_stream || o->reserve_cus != c->opt.reserve_cus) {
    c->err = "se3d_set_options: pair_chunks, hi_priority_stream and reserve_cus are fixed at se3d_create_ex";
    return SE3D_ERR_ARG;
  }
  // the chunk plan was laid out under the creation's options: a form that needs the whole pair table cannot be switched on afterwards
  if (c->plan.m != se3d_pair_chunks_of(*o, c->D.d.N, c->D.d.P, c->D.d.H) && c->plan.m != 0 && se3d_pair_chunks_of(*o, c->D.d.N, c->D.d.P, c->D.d.H) == 0) {
    c->err = "se3d_set_options: this context keeps its pair table in column chunks; pairs_form / sweep_form need a context created with them (or pair_chunks = -1)";
    return SE3D_ERR_ARG;
  }
  c->opt = *o;
  ++c->opt_version;
  return SE3D_OK;
}
int se3d_get_options(se3d_ctx *c, se3d_options *o) {
  if (!c || !o) return SE3D_ERR_ARG;
  *o = c->opt;
  return SE3D_OK;
}

void se3d_destroy(se3d_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->g1.exec) (void)hipGraphExecDestroy(c->g1.exec);
  if (c->g2.exec) (void)hipGraphExecDestroy(c->g2.exec);
  for (void *p : c->allocs) (void)hipFree(p);
  for (se3d_ctx::Slab *sl : {&c->sl_in1, &c->sl_out1, &c->sl_in2, &c->sl_out2})
    if (sl->h) (void)hipHostFree(sl->h);
  if (c->h_t) (void)hipHostFree(c->h_t);
  if (c->h_rp_off) (void)hipHostFree(c->h_rp_off);
  if (c->rpb_cap > 0) { (void)hipFree(c->RB.dense); (void)hipFree(c->RB.valid); (void)hipFree(c->RB.slot); (void)hipFree(c->RB.det_offset); (void)hipFree(c->RB.out); (void)hipFree(c->RB.out_src); }   // rpb_cap > 0 <=> RB owns all six
  for (auto &sl : c->ss) {
    if (sl.h_rag) (void)hipHostFree(sl.h_rag);
    if (sl.h_off) (void)hipHostFree(sl.h_off);
    if (sl.h_fused) (void)hipHostFree(sl.h_fused);
    if (sl.h_pred) (void)hipHostFree(sl.h_pred);
    if (sl.h_out_off) (void)hipHostFree(sl.h_out_off);
    if (sl.h_fb) (void)hipHostFree(sl.h_fb);
    if (sl.in_done) (void)hipEventDestroy(sl.in_done);
    if (sl.compute_done) (void)hipEventDestroy(sl.compute_done);
    if (sl.off_done) (void)hipEventDestroy(sl.off_done);
    if (sl.d2h_start) (void)hipEventDestroy(sl.d2h_start);
  }
  for (auto &set : c->ev2)
    for (auto &e : set)
      if (e) (void)hipEventDestroy(e);
  for (auto &e : c->t_done)
    if (e) (void)hipEventDestroy(e);
  for (auto &set : c->ev)
    for (auto &e : set)
      if (e) (void)hipEventDestroy(e);
  if (c->stream_hi) { (void)hipStreamSynchronize(c->stream_hi); (void)hipStreamDestroy(c->stream_hi); }
  for (auto &e : c->ev_chunk)
    if (e) (void)hipEventDestroy(e);
  for (auto &e : c->ev_hi)
    if (e) (void)hipEventDestroy(e);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

const char *se3d_last_error(const se3d_ctx *c) { return c ? c->err.c_str() : t_create_err.c_str(); }

// Device memory se3d_create(rig with n_cam cameras, limits) allocates, in bytes (host arithmetic only; no device is touched): the
// allocation list of se3d_create walked with a counting allocator.  -1 for sizes se3d_create rejects.
int64_t se3d_device_bytes(int32_t n_cam, const se3d_limits *limits) { return se3d_device_bytes_ex(n_cam, limits, nullptr); }
int64_t se3d_device_bytes_ex(int32_t n_cam, const se3d_limits *limits, const se3d_options *options) {
  if (!limits || !options_valid(options)) return -1;
  const int N = n_cam, B = limits->max_frames, P = limits->max_det_per_cam, H = limits->max_hyp;
  const int T = limits->max_tracks > 0 ? limits->max_tracks : H;
  if (N < 2 || N > SE3D_MAX_CAMS || B < 1 || P < 1 || P > SE3D_MUNKRES_MAX || H < P || H > SE3D_MUNKRES_MAX || T > SE3D_MUNKRES_MAX || (int64_t)B * H >= ((int64_t)1 << 25)) return -1;
  se3d_ctx c;
  c.dry = true;
  c.limits = *limits;
  if (options) c.opt = *options;
  se3d_params prm{};
  RigDev rd;
  (void)alloc_device_state(&c, N, &prm, limits, rd);
  return (int64_t)c.dry_bytes;
}

static int stage_frames(se3d_ctx *c, int32_t n_frames, const se3d_frame_in *frames, bool sync, bool upload = true) {
  if (!c || !frames || n_frames < 1) return SE3D_ERR_ARG;
  const int B = c->D.d.B, N = c->D.d.N, P = c->D.d.P;
  if (n_frames > B) { c->err = "n_frames exceeds limits.max_frames"; return SE3D_ERR_CAPACITY; }
  HIPCHK(c, hipSetDevice(c->device));
  memset(c->h_nraw, 0, sizeof(int32_t) * (size_t)B * N);
  for (int f = 0; f < n_frames; ++f) {
    const se3d_frame_in &fr = frames[f];
    if (fr.n_cam != N || !fr.det_offset || (fr.det_offset[N] > 0 && !fr.persons)) { c->err = "bad frame"; return SE3D_ERR_ARG; }
    for (int cam = 0; cam < N; ++cam) {
      const int n = fr.det_offset[cam + 1] - fr.det_offset[cam];
      if (n < 0) { c->err = "det_offset not monotone"; return SE3D_ERR_ARG; }
      if (n > P) { c->err = "detections per camera exceed limits.max_det_per_cam"; return SE3D_ERR_CAPACITY; }
      c->h_fb[(size_t)f * N + cam] = fr.fb_delay ? fr.fb_delay[cam] : -1.0f;
      c->h_nraw[f * N + cam] = n;
      if (n) memcpy(c->h_raw + ((size_t)(f * N + cam)) * P, fr.persons + fr.det_offset[cam], sizeof(se3d_person2d) * (size_t)n);
    }
  }
  // one copy: {nraw, fb} header and the staged frames' detections (in1 slab, raw last)
  if (upload) HIPCHK(c, hipMemcpyAsync(c->sl_in1.d, c->sl_in1.h, c->off_raw + sizeof(se3d_person2d) * (size_t)n_frames * N * P, hipMemcpyHostToDevice, c->stream));
  if (sync) HIPCHK(c, hipStreamSynchronize(c->stream));   // the pinned slab may be refilled as soon as this returns
  c->staged_frames = n_frames;
  c->ran = false;
  c->ran_dlt = false;
  c->dlt_hyp = 0;
  return SE3D_OK;
}

int se3d_stage_frames(se3d_ctx *c, int32_t n_frames, const se3d_frame_in *frames) { return stage_frames(c, n_frames, frames, true); }

int se3d_run_stage1(se3d_ctx *c) {
  if (!c) return SE3D_ERR_ARG;
  if (c->D.d.shard_world > 1) { c->err = "a hypothesis shard is set (se3d_set_shard, world > 1): use the se3d_run_stage*_shard_pre / _post pairs, the plain run calls would skip the other ranks' hypotheses"; return SE3D_ERR_STATE; }
  if (c->staged_frames < 1) { c->err = "no frames staged"; return SE3D_ERR_STATE; }
  HIPCHK(c, hipSetDevice(c->device));
  Se3dDev D = c->D;
  D.d.B = c->staged_frames;  // grids cover the staged frames only (array strides do not depend on B)
  // the per-frame workspace (assoc, nviews, nhyp, flags, hypP) is reset by se3d_k_normalize, the first kernel of the pass
  se3d_launch_stage1(D, c->opt, c->pair_ab, c->stream, next_ev1(c), 15, c->stream_hi, c->ev_hi, &c->plan, c->ev_chunk);
  HIPCHK(c, hipGetLastError());
  c->ran = true;
  return SE3D_OK;
}

int se3d_run_stage1_phases(se3d_ctx *c, int32_t phases) {
  if (!c || (phases & ~15) || !phases) return SE3D_ERR_ARG;
  if (c->D.d.shard_world > 1) { c->err = "a hypothesis shard is set (se3d_set_shard, world > 1): use the se3d_run_stage*_shard_pre / _post pairs, the plain run calls would skip the other ranks' hypotheses"; return SE3D_ERR_STATE; }
  if (c->staged_frames < 1) { c->err = "no frames staged"; return SE3D_ERR_STATE; }
  HIPCHK(c, hipSetDevice(c->device));
  Se3dDev D = c->D;
  D.d.B = c->staged_frames;
  if (phases & 1) c->cur_ev1 = next_ev1(c);   // one event set per pass, begun by the part that holds phase 1 and completed by the part with phase 4
  se3d_launch_stage1(D, c->opt, c->pair_ab, c->stream, c->cur_ev1, phases, c->stream_hi, c->ev_hi, &c->plan, c->ev_chunk);
  HIPCHK(c, hipGetLastError());
  if (phases & 4) c->ran = true;
  return SE3D_OK;
}

int se3d_sync(se3d_ctx *c) {
  if (!c) return SE3D_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  collect_profile(c);
  return SE3D_OK;
}

void *se3d_stream_handle(se3d_ctx *c) { return c ? reinterpret_cast<void *>(c->stream) : nullptr; }

static int fetch_stage1_impl(se3d_ctx *c, int32_t n_frames, se3d_frame_out *outs, bool download) {
  if (!c || !outs || n_frames < 1) return SE3D_ERR_ARG;
  if (!c->ran || n_frames > c->staged_frames) { c->err = "fetch before run"; return SE3D_ERR_STATE; }
  const Se3dDev &D = c->D;
  const int N = D.d.N, H = D.d.H;
  HIPCHK(c, hipSetDevice(c->device));
  if (!download) {   // the graph of the call already holds the download
  } else if (n_frames == D.d.B) {   // the whole batch: one copy of the out1 slab
    HIPCHK(c, hipMemcpyAsync(c->sl_out1.h, c->sl_out1.d, c->sl_out1.bytes, hipMemcpyDeviceToHost, c->stream));
  } else {
  HIPCHK(c, hipMemcpyAsync(c->h_out, D.outP, sizeof(se3d_person3d) * (size_t)n_frames * H, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->h_count, D.out_count, sizeof(int32_t) * (size_t)n_frames, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->h_nhyp, D.nhyp, sizeof(int32_t) * (size_t)n_frames, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->h_flags, D.flags, sizeof(uint32_t) * (size_t)n_frames, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->h_outhyp, D.out_hyp, sizeof(int32_t) * (size_t)n_frames * H, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->h_outassoc, D.out_assoc, sizeof(int32_t) * (size_t)n_frames * H * N, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  collect_profile(c);
  int rc = SE3D_OK;
  for (int f = 0; f < n_frames; ++f) {
    se3d_frame_out &o = outs[f];
    const int np = c->h_count[f];
    o.n_hyp = c->h_nhyp[f];
    o.flags = c->h_flags[f];
    if (np > o.capacity) { o.n_persons = 0; c->err = "output capacity too small"; rc = SE3D_ERR_CAPACITY; continue; }
    o.n_persons = np;
    if (np && o.persons) memcpy(o.persons, c->h_out + (size_t)f * H, sizeof(se3d_person3d) * (size_t)np);
    if (np && o.person_hyp) memcpy(o.person_hyp, c->h_outhyp + (size_t)f * H, sizeof(int32_t) * (size_t)np);
    if (o.hyp_assoc && o.hyp_capacity > 0) {
      const int nh = o.n_hyp < o.hyp_capacity ? o.n_hyp : o.hyp_capacity;
      if (nh) memcpy(o.hyp_assoc, c->h_outassoc + (size_t)f * H * N, sizeof(int32_t) * (size_t)nh * N);
      if (o.n_hyp > o.hyp_capacity) o.flags |= SE3D_FLAG_HYP_OVERFLOW;
    }
  }
  return rc;
}

int se3d_fetch_stage1(se3d_ctx *c, int32_t n_frames, se3d_frame_out *outs) { return fetch_stage1_impl(c, n_frames, outs, true); }

int se3d_triangulate_batch(se3d_ctx *c, int32_t n_frames, const se3d_frame_in *frames, se3d_frame_out *outs) {
  if (c && n_frames == 1 && graph_wanted(c)) {   // one frame per call: upload, kernel chain and download as one graph replay
    int rc = stage_frames(c, 1, frames, false, /*upload=*/false);
    if (rc != SE3D_OK) return rc;
    Se3dDev D = c->D;
    const int g = graph_replay(c, c->g1, [&] {
      (void)hipMemcpyAsync(c->sl_in1.d, c->sl_in1.h, c->sl_in1.bytes, hipMemcpyHostToDevice, c->stream);
      se3d_launch_stage1(D, c->opt, c->pair_ab, c->stream, nullptr, 15, nullptr, nullptr, &c->plan, nullptr);   // (at capture; a replay runs the same kernels)
      (void)hipMemcpyAsync(c->sl_out1.h, c->sl_out1.d, c->sl_out1.bytes, hipMemcpyDeviceToHost, c->stream);
    });
    if (g < 0) return g;
    if (g == 1) {
      c->ran = true;
      return fetch_stage1_impl(c, 1, outs, /*download=*/false);
    }
    // capture not possible on this runtime: the direct path below (the slab is filled; upload it)
    HIPCHK(c, hipMemcpyAsync(c->sl_in1.d, c->sl_in1.h, c->sl_in1.bytes, hipMemcpyHostToDevice, c->stream));
    rc = se3d_run_stage1(c);
    if (rc != SE3D_OK) return rc;
    return se3d_fetch_stage1(c, 1, outs);
  }
  int rc = stage_frames(c, n_frames, frames, false);   // no wait after the upload: the fetch below synchronises the stream
  if (rc != SE3D_OK) return rc;
  rc = se3d_run_stage1(c);
  if (rc != SE3D_OK) return rc;
  return se3d_fetch_stage1(c, n_frames, outs);
}

int se3d_triangulate(se3d_ctx *c, const se3d_frame_in *frame, se3d_frame_out *out) { return se3d_triangulate_batch(c, 1, frame, out); }

// ---- config 2: batched DLT triangulation + UT covariance with the association given --------------------------------
int se3d_stage_assoc(se3d_ctx *c, int32_t n_hyp, const int32_t *assoc) {
  if (!c || !assoc || n_hyp < 1) return SE3D_ERR_ARG;
  if (c->staged_frames < 1) { c->err = "se3d_stage_assoc: no frames staged"; return SE3D_ERR_STATE; }
  if (n_hyp > c->D.d.H) { c->err = "se3d_stage_assoc: n_hyp exceeds limits.max_hyp"; return SE3D_ERR_CAPACITY; }
  HIPCHK(c, hipSetDevice(c->device));
  const size_t n = (size_t)c->staged_frames * n_hyp * c->D.d.N;
  memcpy(c->h_outassoc, assoc, sizeof(int32_t) * n);  // pinned staging ([B][H][N] ints, shared with se3d_fetch_stage1)
  HIPCHK(c, hipMemcpyAsync(c->d_assoc_in, c->h_outassoc, sizeof(int32_t) * n, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  c->dlt_hyp = n_hyp;
  c->ran_dlt = false;
  return SE3D_OK;
}

int se3d_run_dlt(se3d_ctx *c) {
  if (!c) return SE3D_ERR_ARG;
  if (c->staged_frames < 1 || c->dlt_hyp < 1) { c->err = "se3d_run_dlt needs se3d_stage_frames + se3d_stage_assoc"; return SE3D_ERR_STATE; }
  HIPCHK(c, hipSetDevice(c->device));
  Se3dDev D = c->D;
  D.d.B = c->staged_frames;
  hipEvent_t *ev = next_ev1(c);
  if (ev) c->ev_dlt[(c->ev_head + se3d_ctx::kRing - 1) % se3d_ctx::kRing] = true;
  se3d_launch_dlt(D, c->opt, c->d_assoc_in, c->dlt_hyp, c->stream, ev);
  HIPCHK(c, hipGetLastError());
  c->ran_dlt = true;
  c->ran = false;  // the association workspace now holds the caller's table, not a sweep result
  return SE3D_OK;
}

int se3d_fetch_dlt(se3d_ctx *c, int32_t n_frames, se3d_person3d *persons_out) {
  if (!c || !persons_out || n_frames < 1) return SE3D_ERR_ARG;
  if (!c->ran_dlt || n_frames > c->staged_frames) { c->err = "se3d_fetch_dlt before se3d_run_dlt"; return SE3D_ERR_STATE; }
  HIPCHK(c, hipSetDevice(c->device));
  const size_t H = c->D.d.H, nh = c->dlt_hyp;
  HIPCHK(c, hipMemcpy2DAsync(c->h_out, sizeof(se3d_person3d) * nh, c->D.hypP, sizeof(se3d_person3d) * H, sizeof(se3d_person3d) * nh, (size_t)n_frames,
                             hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  collect_profile(c);
  memcpy(persons_out, c->h_out, sizeof(se3d_person3d) * nh * (size_t)n_frames);
  return SE3D_OK;
}

int se3d_triangulate_assoc(se3d_ctx *c, int32_t n_frames, const se3d_frame_in *frames, int32_t n_hyp, const int32_t *assoc, se3d_person3d *persons_out) {
  int rc = se3d_stage_frames(c, n_frames, frames);
  if (rc != SE3D_OK) return rc;
  if ((rc = se3d_stage_assoc(c, n_hyp, assoc)) != SE3D_OK) return rc;
  if ((rc = se3d_run_dlt(c)) != SE3D_OK) return rc;
  return se3d_fetch_dlt(c, n_frames, persons_out);
}

int se3d_pack_stage1_device(se3d_ctx *c, void *d_persons, void *d_counts) {
  if (!c || !d_persons || !d_counts) return SE3D_ERR_ARG;
  if (!c->ran) return SE3D_ERR_STATE;
  const Se3dDev &D = c->D;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(d_persons, D.outP, sizeof(se3d_person3d) * (size_t)c->staged_frames * D.d.H, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_counts, D.out_count, sizeof(int32_t) * (size_t)c->staged_frames, hipMemcpyDeviceToDevice, c->stream));
  return SE3D_OK;  // asynchronous on the context stream: se3d_sync before another stream / the host reads the buffers
}

int se3d_hungarian(se3d_ctx *c, int32_t *assignment, double *cost, const double *dist, int32_t nR, int32_t nC) {
  if (!c || !assignment || !dist || nR < 0 || nC < 0) return SE3D_ERR_ARG;
  if (nR > SE3D_MUNKRES_MAX || nC > SE3D_MUNKRES_MAX) { c->err = "se3d_hungarian: at most 256 rows/cols"; return SE3D_ERR_CAPACITY; }
  if (cost) *cost = 0.0;
  for (int r = 0; r < nR; ++r) assignment[r] = -1;
  if (nR == 0 || nC == 0) return SE3D_OK;
  HIPCHK(c, hipSetDevice(c->device));
  double *d_dist = c->d_hung;           // preallocated in the context (no allocation per call, nothing to leak on an error return)
  int *d_assign = c->d_hung_assign, *d_status = d_assign + nR;
  HIPCHK(c, hipMemcpyAsync(d_dist, dist, sizeof(double) * (size_t)nR * nC, hipMemcpyHostToDevice, c->stream));
  se3d_launch_munkres(d_dist, d_assign, d_status, nR, nC, c->stream, c->opt.munkres_form);   // parity tests force the block / the one-wavefront form
  std::vector<int> h(nR + 1);
  HIPCHK(c, hipMemcpyAsync(h.data(), d_assign, sizeof(int) * (size_t)(nR + 1), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (h[nR] != 0) { c->err = "se3d_hungarian: iteration guard fired"; return SE3D_ERR_DEVICE; }
  for (int r = 0; r < nR; ++r) {
    assignment[r] = h[r];
    if (cost && h[r] >= 0) *cost += dist[r + (size_t)nR * h[r]];  // computeassignmentcost (HUN:208-218)
  }
  return SE3D_OK;
}

int se3d_get_pair_costs(se3d_ctx *c, int32_t frame, int32_t cam_a, int32_t cam_b, double *out, int32_t *n_a, int32_t *n_b) {
  if (!c || !out) return SE3D_ERR_ARG;
  const Se3dDev &D = c->D;
  if (!c->ran || frame < 0 || frame >= c->staged_frames) return SE3D_ERR_STATE;
  if (cam_a < 0 || cam_b <= cam_a || cam_b >= D.d.N) return SE3D_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  const int P = D.d.P;
  size_t off = ((size_t)frame * D.d.npairs + se3d_pair_index(cam_a, cam_b, D.d.N)) * P * P;
  if (D.d.pair_chunks) {
    // column chunks: the table of the pass is gone (every chunk buffer has been overwritten since); the block is computed again by the
    // pair kernel of the pass into slot 0 of the frame (one-entry pair list in the first element of C, free between passes)
    // (both are scratch between passes — the next se3d_run_stage1 rewrites them — and the slot is cleared to -1 first, "no joint shared":
    // the kernel writes nothing for a camera without valid detections)
    const int2 one = make_int2(cam_a, cam_b);
    HIPCHK(c, hipMemcpyAsync(D.C, &one, sizeof(int2), hipMemcpyHostToDevice, c->stream));
    off = (size_t)frame * D.d.pair_stride * P * P;
    {
      std::vector<double> neg((size_t)P * P, -1.0);
      HIPCHK(c, hipMemcpyAsync(D.pair + off, neg.data(), sizeof(double) * (size_t)P * P, hipMemcpyHostToDevice, c->stream));
      HIPCHK(c, hipStreamSynchronize(c->stream));   // `neg` leaves scope
    }
    Se3dDev D1 = D;
    D1.d.B = 1;
    se3d_launch_pair_block(D1, reinterpret_cast<const int2 *>(D.C), frame, c->stream);
    HIPCHK(c, hipGetLastError());
  }
  int32_t nv[2];
  HIPCHK(c, hipMemcpyAsync(out, D.pair + off, sizeof(double) * (size_t)P * P, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&nv[0], D.nvalid + frame * D.d.N + cam_a, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(&nv[1], D.nvalid + frame * D.d.N + cam_b, sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  if (n_a) *n_a = nv[0];
  if (n_b) *n_b = nv[1];
  return SE3D_OK;
}

// ---------------- stage 2 ----------------
int se3d_refine_reset(se3d_ctx *c) {
  if (!c) return SE3D_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  collect_profile(c);
  return stage2_reset(c);
}

static int fetch_stage2_range(se3d_ctx *c, int seq0, int n_seq, se3d_fused_out *outs, bool download = true) {
  const Se3dStage2 &Q = c->Q;
  const size_t H = Q.H;
  if (!download) {   // the graph of the call already holds the download
  } else if (seq0 == 0 && n_seq == Q.S) {   // all sequences: one copy of the out2 slab
    HIPCHK(c, hipMemcpyAsync(c->sl_out2.h, c->sl_out2.d, c->sl_out2.bytes, hipMemcpyDeviceToHost, c->stream));
  } else {
  HIPCHK(c, hipMemcpyAsync(c->h_fused + seq0 * H, Q.fused + seq0 * H, sizeof(se3d_person3d) * n_seq * H, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->h_pred + seq0 * H, Q.pred + seq0 * H, sizeof(se3d_person3d) * n_seq * H, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->h_iters + seq0 * H, Q.iters + seq0 * H, sizeof(int32_t) * n_seq * H, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->h_obj + seq0 * H, Q.obj + seq0 * H, sizeof(double) * n_seq * H, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->h_nout + seq0, Q.n_out + seq0, sizeof(int32_t) * n_seq, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->h_ntracks + seq0, Q.n_tracks + seq0, sizeof(int32_t) * n_seq, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->h_flags2 + seq0, Q.flags2 + seq0, sizeof(uint32_t) * n_seq, hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipMemcpyAsync(c->h_preddt + seq0, Q.pred_dt + seq0, sizeof(double) * n_seq, hipMemcpyDeviceToHost, c->stream));
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  collect_profile(c);
  int rc = SE3D_OK;
  for (int k = 0; k < n_seq; ++k) {
    se3d_fused_out &o = outs[k];
    const int s = seq0 + k;
    const int np = c->h_nout[s];
    o.n_tracks = c->h_ntracks[s];
    o.flags = c->h_flags2[s];
    o.pred_delta_t = (float)c->h_preddt[s];
    if (np > o.capacity) { o.n_persons = 0; c->err = "output capacity too small"; rc = SE3D_ERR_CAPACITY; continue; }
    o.n_persons = np;
    if (np && o.fused) memcpy(o.fused, c->h_fused + s * H, sizeof(se3d_person3d) * (size_t)np);
    if (np && o.fused_pred) memcpy(o.fused_pred, c->h_pred + s * H, sizeof(se3d_person3d) * (size_t)np);
    if (np && o.lm_iterations) memcpy(o.lm_iterations, c->h_iters + s * H, sizeof(int32_t) * (size_t)np);
    if (np && o.lm_objective) memcpy(o.lm_objective, c->h_obj + s * H, sizeof(double) * (size_t)np);
  }
  return rc;
}

int se3d_refine(se3d_ctx *c, int32_t seq, int32_t n_det, const se3d_person3d *dets, int32_t n_cam, const float *fb_delay_per_cam, double t,
                se3d_fused_out *out) {
  if (!c || !out || n_det < 0 || (n_det > 0 && !dets)) return SE3D_ERR_ARG;
  Se3dStage2 Q = c->Q;
  if (seq < 0 || seq >= Q.S) { c->err = "sequence index out of range"; return SE3D_ERR_ARG; }
  if (n_det > Q.H) { c->err = "detections exceed limits.max_hyp"; return SE3D_ERR_CAPACITY; }
  HIPCHK(c, hipSetDevice(c->device));
  collect_profile(c);
  const size_t H = Q.H, N = Q.N;
  // every return path of this call synchronises the stream, so the pinned in2 slab is free to refill here
  for (size_t i = 0; i < N; ++i) c->h_fb_ref[seq * N + i] = (fb_delay_per_cam && (int)i < n_cam) ? fb_delay_per_cam[i] : -1.0f;
  c->h_t_ref[seq] = t;
  c->h_ndet2[seq] = n_det;
  if (n_det) memcpy(c->h_in2 + seq * H, dets, sizeof(se3d_person3d) * (size_t)n_det);
  if (Q.S == 1 && graph_wanted(c)) {   // one frame per call: upload, the step's kernel(s) and download as one graph replay
    const int g = graph_replay(c, c->g2, [&] {
      (void)hipMemcpyAsync(c->sl_in2.d, c->sl_in2.h, c->sl_in2.bytes, hipMemcpyHostToDevice, c->stream);
      Se3dStage2 Qg = c->Q;
      Qg.dets = c->d_in2; Qg.n_det = c->d_ndet2; Qg.fb_delay = c->d_fb_ref; Qg.t = c->d_t;
      bool walked = false;
      if (!c->opt.refine_steps) {
        Se3dWalk Wk{};
        Wk.n_steps = 1; Wk.slot0[0] = 0; Wk.t = c->d_t; Wk.t_stride = 1;
        Wk.outP = c->d_in2; Wk.out_count = c->d_ndet2; Wk.fb = c->d_fb_ref;
        Wk.Wf = c->Q.fused; Wk.Wp = c->Q.pred; Wk.Wit = c->Q.iters; Wk.Wobj = c->Q.obj; Wk.Wn = c->Q.n_out; Wk.Wdt = c->Q.pred_dt;
        walked = se3d_launch_stage2_walk(c->Q, c->opt, Wk, 1, c->stream);
      }
      if (!walked) se3d_launch_stage2(Qg, c->opt, 0, 1, c->stream, nullptr, 3);
      (void)hipMemcpyAsync(c->sl_out2.h, c->sl_out2.d, c->sl_out2.bytes, hipMemcpyDeviceToHost, c->stream);
    });
    if (g < 0) return g;
    if (g == 1) {
      c->ran2 = true;
      return fetch_stage2_range(c, 0, 1, out, /*download=*/false);
    }
  }
  if (Q.S == 1) {   // one copy: {n_det, t, fb_delay} header and the n_det measurements (in2 slab, measurements last)
    HIPCHK(c, hipMemcpyAsync(c->sl_in2.d, c->sl_in2.h, c->off_in2 + sizeof(se3d_person3d) * (size_t)n_det, hipMemcpyHostToDevice, c->stream));
  } else {
    if (n_det) HIPCHK(c, hipMemcpyAsync(c->d_in2 + seq * H, c->h_in2 + seq * H, sizeof(se3d_person3d) * (size_t)n_det, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_ndet2 + seq, c->h_ndet2 + seq, sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_fb_ref + seq * N, c->h_fb_ref + seq * N, sizeof(float) * N, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->d_t + seq, c->h_t_ref + seq, sizeof(double), hipMemcpyHostToDevice, c->stream));
  }
  Q.dets = c->d_in2; Q.n_det = c->d_ndet2; Q.fb_delay = c->d_fb_ref; Q.t = c->d_t;
  // One frame per call on a one-sequence context (the reference's operating point, PP:505): the step as ONE launch of the fused walk kernel
  // (association, LM with three persons per wavefront, lifecycle; n_steps = 1) instead of five dependent launches — round 4, latency_B1.
  // options.refine_steps = 1 keeps the per-step kernels.
  bool walked = false;
  if (Q.S == 1) {
    if (!c->opt.refine_steps) {
      Se3dWalk Wk{};
      Wk.n_steps = 1;
      Wk.slot0[0] = 0;
      Wk.t = c->d_t; Wk.t_stride = 1;
      Wk.outP = c->d_in2; Wk.out_count = c->d_ndet2; Wk.fb = c->d_fb_ref;
      Wk.Wf = c->Q.fused; Wk.Wp = c->Q.pred; Wk.Wit = c->Q.iters; Wk.Wobj = c->Q.obj; Wk.Wn = c->Q.n_out; Wk.Wdt = c->Q.pred_dt;
      hipEvent_t *ev = next_ev2(c);
      if (ev) { c->ev2_walk[(c->ev2_head + se3d_ctx::kRing - 1) % se3d_ctx::kRing] = true; (void)hipEventRecord(ev[0], c->stream); }
      walked = se3d_launch_stage2_walk(c->Q, c->opt, Wk, 1, c->stream);
      if (walked && ev) (void)hipEventRecord(ev[3], c->stream);
      if (!walked && ev) {   // not applicable (limits too large for the kernel's LDS): give the event set back
        c->ev2_walk[(c->ev2_head + se3d_ctx::kRing - 1) % se3d_ctx::kRing] = false;
        c->ev2_head = (c->ev2_head + se3d_ctx::kRing - 1) % se3d_ctx::kRing;
        --c->ev2_pending;
      }
    }
  }
  if (!walked) se3d_launch_stage2(Q, c->opt, seq, 1, c->stream, next_ev2(c), 3);
  HIPCHK(c, hipGetLastError());
  c->ran2 = true;
  return fetch_stage2_range(c, seq, 1, out);
}

// ---- sequence state of pose_prior (g_tracks and the node globals, PP:54-60, 68-76, 123) out of / into a context: a host blob that does
// not depend on the context's capacity, so that a host layer can move a running sequence into a context with larger limits (the
// reference's vectors grow without bound, PP:570-580: g_tracks.push_back) ----
extern "C++" {
namespace {
struct SeqStateHeader { uint32_t magic; int32_t n_tracks, next_id, frame_nr; double t_prev, fb_buf[3]; uint32_t flags2, pad; };
struct SeqStateTrack {
  int32_t id, num_obs; uint32_t has, pad; double t_prev, height_prev, root_prev[3], prev[SE3D_NFU * 3], vel[SE3D_NFU * 9];
};
constexpr uint32_t kSeqMagic = 0x53325354u;  // "S2ST"
template <typename T> bool d2h(se3d_ctx *c, T *dst, const T *src, size_t n) { return hip_ok(c, hipMemcpy(dst, src, n * sizeof(T), hipMemcpyDeviceToHost), "hipMemcpy D2H"); }
template <typename T> bool h2d(se3d_ctx *c, T *dst, const T *src, size_t n) { return hip_ok(c, hipMemcpy(dst, src, n * sizeof(T), hipMemcpyHostToDevice), "hipMemcpy H2D"); }
}  // namespace
}  // extern "C++"

int64_t se3d_refine_state_bytes(const se3d_ctx *c) {
  return c ? (int64_t)(sizeof(SeqStateHeader) + (size_t)c->Q.T * sizeof(SeqStateTrack)) : 0;
}

int se3d_refine_save(se3d_ctx *c, int32_t seq, void *blob, int64_t capacity_bytes, int64_t *bytes) {
  if (!c || !blob || !bytes) return SE3D_ERR_ARG;
  const Se3dStage2 &Q = c->Q;
  if (seq < 0 || seq >= Q.S) { c->err = "sequence index out of range"; return SE3D_ERR_ARG; }
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  const size_t T = Q.T, s0 = (size_t)seq * T;
  SeqStateHeader h{};
  h.magic = kSeqMagic;
  if (!d2h(c, &h.n_tracks, Q.n_tracks + seq, 1) || !d2h(c, &h.next_id, Q.next_id + seq, 1) || !d2h(c, &h.frame_nr, Q.frame_nr + seq, 1) ||
      !d2h(c, &h.t_prev, Q.t_prev + seq, 1) || !d2h(c, h.fb_buf, Q.fb_buf + (size_t)seq * 3, 3) || !d2h(c, &h.flags2, Q.flags2 + seq, 1))
    return SE3D_ERR_DEVICE;
  if (h.n_tracks < 0 || (size_t)h.n_tracks > T) { c->err = "se3d_refine_save: corrupt track count"; return SE3D_ERR_STATE; }
  const int64_t need = (int64_t)(sizeof(SeqStateHeader) + (size_t)h.n_tracks * sizeof(SeqStateTrack));
  *bytes = need;
  if (capacity_bytes < need) { c->err = "se3d_refine_save: blob too small (se3d_refine_state_bytes)"; return SE3D_ERR_CAPACITY; }
  std::vector<int32_t> order(T), id(T), nobs(T);
  std::vector<uint32_t> has(T);
  std::vector<double> tp(T), hp(T), rp(T * 3), prev(T * SE3D_NFU * 3), vel(T * SE3D_NFU * 9);
  if (!d2h(c, order.data(), Q.order + s0, T) || !d2h(c, id.data(), Q.id + s0, T) || !d2h(c, nobs.data(), Q.num_obs + s0, T) || !d2h(c, has.data(), Q.has + s0, T) ||
      !d2h(c, tp.data(), Q.trk_t_prev + s0, T) || !d2h(c, hp.data(), Q.height_prev + s0, T) || !d2h(c, rp.data(), Q.root_prev + s0 * 3, T * 3) ||
      !d2h(c, prev.data(), Q.prev + s0 * SE3D_NFU * 3, T * SE3D_NFU * 3) || !d2h(c, vel.data(), Q.vel + s0 * SE3D_NFU * 9, T * SE3D_NFU * 9))
    return SE3D_ERR_DEVICE;
  unsigned char *out = static_cast<unsigned char *>(blob);
  memcpy(out, &h, sizeof h);
  for (int k = 0; k < h.n_tracks; ++k) {   // g_tracks vector order
    const size_t sl = (size_t)order[k];
    if (sl >= T) { c->err = "se3d_refine_save: corrupt track order"; return SE3D_ERR_STATE; }
    SeqStateTrack t{};
    t.id = id[sl]; t.num_obs = nobs[sl]; t.has = has[sl]; t.t_prev = tp[sl]; t.height_prev = hp[sl];
    memcpy(t.root_prev, &rp[sl * 3], sizeof t.root_prev);
    memcpy(t.prev, &prev[sl * SE3D_NFU * 3], sizeof t.prev);
    memcpy(t.vel, &vel[sl * SE3D_NFU * 9], sizeof t.vel);
    for (int j = 0; j < SE3D_NFU; ++j)   // joints outside prevEstimate: their device slots are never read and hold whatever was there — zeros in the blob
      if (!((t.has >> j) & 1u)) {
        for (int a = 0; a < 3; ++a) t.prev[j * 3 + a] = 0.0;
        for (int a = 0; a < 9; ++a) t.vel[j * 9 + a] = 0.0;
      }
    memcpy(out + sizeof h + (size_t)k * sizeof t, &t, sizeof t);
  }
  return SE3D_OK;
}

int se3d_refine_load(se3d_ctx *c, int32_t seq, const void *blob, int64_t bytes) {
  if (!c || !blob || bytes < (int64_t)sizeof(SeqStateHeader)) return SE3D_ERR_ARG;
  const Se3dStage2 &Q = c->Q;
  if (seq < 0 || seq >= Q.S) { c->err = "sequence index out of range"; return SE3D_ERR_ARG; }
  SeqStateHeader h;
  memcpy(&h, blob, sizeof h);
  if (h.magic != kSeqMagic || h.n_tracks < 0 || bytes < (int64_t)(sizeof h + (size_t)h.n_tracks * sizeof(SeqStateTrack))) { c->err = "se3d_refine_load: not a sequence state"; return SE3D_ERR_ARG; }
  const size_t T = Q.T, s0 = (size_t)seq * T;
  if ((size_t)h.n_tracks > T) { c->err = "se3d_refine_load: more tracks than limits.max_tracks"; return SE3D_ERR_CAPACITY; }
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::vector<int32_t> order(T, 0), id(T, 0), nobs(T, 0);
  std::vector<uint32_t> has(T, 0u);
  std::vector<uint8_t> used(T, 0);
  std::vector<double> tp(T, 0.0), hp(T, 0.0), rp(T * 3, 0.0), prev(T * SE3D_NFU * 3, 0.0), vel(T * SE3D_NFU * 9, 0.0);
  const unsigned char *in = static_cast<const unsigned char *>(blob) + sizeof h;
  for (int k = 0; k < h.n_tracks; ++k) {   // track k of the vector goes to slot k
    SeqStateTrack t;
    memcpy(&t, in + (size_t)k * sizeof t, sizeof t);
    order[k] = k; used[k] = 1; id[k] = t.id; nobs[k] = t.num_obs; has[k] = t.has; tp[k] = t.t_prev; hp[k] = t.height_prev;
    memcpy(&rp[(size_t)k * 3], t.root_prev, sizeof t.root_prev);
    memcpy(&prev[(size_t)k * SE3D_NFU * 3], t.prev, sizeof t.prev);
    memcpy(&vel[(size_t)k * SE3D_NFU * 9], t.vel, sizeof t.vel);
  }
  const uint32_t fl = h.flags2 & ~SE3D_S2FLAG_TRACK_OVERFLOW;   // the overflow flag is a statement about the context the state came from
  if (!h2d(c, Q.n_tracks + seq, &h.n_tracks, 1) || !h2d(c, Q.next_id + seq, &h.next_id, 1) || !h2d(c, Q.frame_nr + seq, &h.frame_nr, 1) ||
      !h2d(c, Q.t_prev + seq, &h.t_prev, 1) || !h2d(c, Q.fb_buf + (size_t)seq * 3, h.fb_buf, 3) || !h2d(c, Q.flags2 + seq, &fl, 1) ||
      !h2d(c, Q.order + s0, order.data(), T) || !h2d(c, Q.used + s0, used.data(), T) || !h2d(c, Q.id + s0, id.data(), T) ||
      !h2d(c, Q.num_obs + s0, nobs.data(), T) || !h2d(c, Q.has + s0, has.data(), T) || !h2d(c, Q.trk_t_prev + s0, tp.data(), T) ||
      !h2d(c, Q.height_prev + s0, hp.data(), T) || !h2d(c, Q.root_prev + s0 * 3, rp.data(), T * 3) ||
      !h2d(c, Q.prev + s0 * SE3D_NFU * 3, prev.data(), T * SE3D_NFU * 3) || !h2d(c, Q.vel + s0 * SE3D_NFU * 9, vel.data(), T * SE3D_NFU * 9))
    return SE3D_ERR_DEVICE;
  return SE3D_OK;
}

int se3d_run_stage2(se3d_ctx *c, int32_t n_seq, int32_t frame_slot0, const double *t) {
  if (!c || !t || n_seq < 1 || frame_slot0 < 0) return SE3D_ERR_ARG;
  if (c->D.d.shard_world > 1) { c->err = "a hypothesis shard is set (se3d_set_shard, world > 1): use the se3d_run_stage*_shard_pre / _post pairs, the plain run calls would skip the other ranks' hypotheses"; return SE3D_ERR_STATE; }
  if (!c->ran || frame_slot0 + n_seq > c->staged_frames) { c->err = "se3d_run_stage2 needs a preceding se3d_run_stage1 over >= n_seq frames"; return SE3D_ERR_STATE; }
  HIPCHK(c, hipSetDevice(c->device));
  Se3dStage2 Q = c->Q;
  double *ht = c->h_t + (size_t)c->t_head * Q.S;
  (void)hipEventSynchronize(c->t_done[c->t_head]);
  memcpy(ht, t, sizeof(double) * (size_t)n_seq);
  HIPCHK(c, hipMemcpyAsync(c->d_t, ht, sizeof(double) * (size_t)n_seq, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipEventRecord(c->t_done[c->t_head], c->stream));
  c->t_head = (c->t_head + 1) % se3d_ctx::kRing;
  Q.dets = c->D.outP + (size_t)frame_slot0 * Q.H; Q.n_det = c->D.out_count + frame_slot0; Q.fb_delay = c->d_fb + (size_t)frame_slot0 * Q.N; Q.t = c->d_t;
  se3d_launch_stage2(Q, c->opt, 0, n_seq, c->stream, next_ev2(c), 3);
  HIPCHK(c, hipGetLastError());
  c->ran2 = true;
  return SE3D_OK;
}

int se3d_run_stage2_walk(se3d_ctx *c, int32_t n_seq, int32_t n_steps, const int32_t *frame_slot0, const double *t) {
  if (!c || !t || !frame_slot0 || n_seq < 1 || n_steps < 1 || n_steps > SE3D_WALK_MAX_STEPS) return SE3D_ERR_ARG;
  if (c->D.d.shard_world > 1) { c->err = "a hypothesis shard is set (se3d_set_shard, world > 1): use the se3d_run_stage*_shard_pre / _post pairs, the plain run calls would skip the other ranks' hypotheses"; return SE3D_ERR_STATE; }
  if (!c->ran) { c->err = "se3d_run_stage2_walk needs a preceding se3d_run_stage1"; return SE3D_ERR_STATE; }
  if ((size_t)n_steps * n_seq > (size_t)c->Q.S) { c->err = "se3d_run_stage2_walk: n_steps * n_seq exceeds limits.max_frames"; return SE3D_ERR_CAPACITY; }
  for (int k = 0; k < n_steps; ++k)
    if (frame_slot0[k] < 0 || frame_slot0[k] + n_seq > c->staged_frames) { c->err = "se3d_run_stage2_walk: frame slot outside the staged batch"; return SE3D_ERR_ARG; }
  HIPCHK(c, hipSetDevice(c->device));
  const Se3dStage2 &Q0 = c->Q;
  double *ht = c->h_t + (size_t)c->t_head * Q0.S;
  (void)hipEventSynchronize(c->t_done[c->t_head]);
  memcpy(ht, t, sizeof(double) * (size_t)n_seq * n_steps);
  HIPCHK(c, hipMemcpyAsync(c->d_t, ht, sizeof(double) * (size_t)n_seq * n_steps, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipEventRecord(c->t_done[c->t_head], c->stream));
  c->t_head = (c->t_head + 1) % se3d_ctx::kRing;
  Se3dWalk Wk{};
  Wk.n_steps = n_steps;
  for (int k = 0; k < n_steps; ++k) Wk.slot0[k] = frame_slot0[k];
  Wk.t = c->d_t; Wk.t_stride = n_seq;
  Wk.outP = c->D.outP; Wk.out_count = c->D.out_count; Wk.fb = c->d_fb_walk ? c->d_fb_walk : c->d_fb;
  Wk.Wf = Q0.fused; Wk.Wp = Q0.pred; Wk.Wit = Q0.iters; Wk.Wobj = Q0.obj; Wk.Wn = Q0.n_out; Wk.Wdt = Q0.pred_dt;
  hipEvent_t *ev = next_ev2(c);
  if (ev) { c->ev2_walk[(c->ev2_head + se3d_ctx::kRing - 1) % se3d_ctx::kRing] = true; (void)hipEventRecord(ev[0], c->stream); }
  if (!se3d_launch_stage2_walk(Q0, c->opt, Wk, n_seq, c->stream)) {
    hipStream_t s2 = c->stream;
    if (c->stream_hi) {  // the chain of 5 x n_steps small launches on the high-priority stream
      (void)hipEventRecord(c->ev_hi[2], c->stream);
      (void)hipStreamWaitEvent(c->stream_hi, c->ev_hi[2], 0);
      s2 = c->stream_hi;
    }
    for (int k = 0; k < n_steps; ++k) {  // per-step kernels with the same per-step views (large association problems)
      const size_t f0 = (size_t)frame_slot0[k];
      Se3dStage2 Q = Q0;
      Q.dets = Wk.outP + f0 * Q.H; Q.n_det = Wk.out_count + f0; Q.fb_delay = Wk.fb + f0 * Q.N; Q.t = c->d_t + (size_t)k * n_seq;
      Q.fused = Q0.fused + f0 * Q.H; Q.pred = Q0.pred + f0 * Q.H; Q.iters = Q0.iters + f0 * Q.H; Q.obj = Q0.obj + f0 * Q.H; Q.n_out = Q0.n_out + f0; Q.pred_dt = Q0.pred_dt + f0;
      se3d_launch_stage2(Q, c->opt, 0, n_seq, s2, nullptr, 3);
    }
    if (s2 != c->stream) {
      (void)hipEventRecord(c->ev_hi[3], s2);
      (void)hipStreamWaitEvent(c->stream, c->ev_hi[3], 0);
    }
  }
  if (ev) (void)hipEventRecord(ev[3], c->stream);
  HIPCHK(c, hipGetLastError());
  c->ran2 = true;
  return SE3D_OK;
}

int se3d_fetch_stage2_walk(se3d_ctx *c, int32_t frame_slot0, int32_t n_seq, se3d_fused_out *outs) {
  if (!c || !outs || n_seq < 1 || frame_slot0 < 0 || frame_slot0 + n_seq > c->Q.S) return SE3D_ERR_ARG;
  if (!c->ran2) { c->err = "fetch before run"; return SE3D_ERR_STATE; }
  HIPCHK(c, hipSetDevice(c->device));
  const int rc = fetch_stage2_range(c, frame_slot0, n_seq, outs);
  if (frame_slot0 > 0) {  // n_tracks and the flags are per-sequence state (current value), not per frame slot
    HIPCHK(c, hipMemcpy(c->h_ntracks, c->Q.n_tracks, sizeof(int32_t) * (size_t)n_seq, hipMemcpyDeviceToHost));
    HIPCHK(c, hipMemcpy(c->h_flags2, c->Q.flags2, sizeof(uint32_t) * (size_t)n_seq, hipMemcpyDeviceToHost));
    for (int k = 0; k < n_seq; ++k) { outs[k].n_tracks = c->h_ntracks[k]; outs[k].flags = c->h_flags2[k]; }
  }
  return rc;
}

int se3d_pack_stage2_walk_device(se3d_ctx *c, int32_t n_frames, void *d_persons, void *d_counts) {
  if (!c || !d_persons || !d_counts || n_frames < 1 || n_frames > c->Q.S) return SE3D_ERR_ARG;
  if (!c->ran2) return SE3D_ERR_STATE;
  const Se3dStage2 &Q = c->Q;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(d_persons, Q.fused, sizeof(se3d_person3d) * (size_t)n_frames * Q.H, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_counts, Q.n_out, sizeof(int32_t) * (size_t)n_frames, hipMemcpyDeviceToDevice, c->stream));
  return SE3D_OK;
}

int se3d_pack_stage2_walk_device_slots(se3d_ctx *c, int32_t n_frames, int32_t slots, void *d_persons, void *d_counts) {
  if (!c || !d_persons || !d_counts || n_frames < 1 || n_frames > c->Q.S || slots < 1 || slots > c->Q.H) return SE3D_ERR_ARG;
  if (!c->ran2) return SE3D_ERR_STATE;
  const Se3dStage2 &Q = c->Q;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpy2DAsync(d_persons, sizeof(se3d_person3d) * (size_t)slots, Q.fused, sizeof(se3d_person3d) * (size_t)Q.H,
                             sizeof(se3d_person3d) * (size_t)slots, (size_t)n_frames, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_counts, Q.n_out, sizeof(int32_t) * (size_t)n_frames, hipMemcpyDeviceToDevice, c->stream));
  return SE3D_OK;
}

int se3d_pack_stage2_walk_device_dense(se3d_ctx *c, int32_t n_frames, int32_t capacity, void *d_persons, void *d_offsets) {
  if (!c || !d_persons || !d_offsets || n_frames < 1 || n_frames > c->Q.S || capacity < 1) return SE3D_ERR_ARG;
  if (!c->ran2) return SE3D_ERR_STATE;
  const Se3dStage2 &Q = c->Q;
  HIPCHK(c, hipSetDevice(c->device));
  se3d_launch_pack_dense(Q.fused, Q.n_out, n_frames, Q.H, capacity, static_cast<int32_t *>(d_offsets), static_cast<se3d_person3d *>(d_persons), c->stream);
  HIPCHK(c, hipGetLastError());
  return SE3D_OK;
}

// ---- hypothesis shard (SURVEY 8e / north_star config 5): replicated association, hypotheses and detections split over the ranks ----
int se3d_set_shard(se3d_ctx *c, int32_t rank, int32_t world) {
  if (!c || world < 1 || rank < 0 || rank >= world) return SE3D_ERR_ARG;
  c->D.d.shard_rank = rank; c->D.d.shard_world = world;
  c->Q.shard_rank = rank; c->Q.shard_world = world;
  return SE3D_OK;
}
int64_t se3d_shard_bytes(se3d_ctx *c, int32_t stage, int32_t n) {
  if (!c || n < 0) return 0;
  const int64_t Hs = (c->D.d.H + c->D.d.shard_world - 1) / c->D.d.shard_world;
  return (int64_t)n * Hs * (int64_t)(stage == 1 ? sizeof(se3d_person3d) : sizeof(Se3dShardRec));
}
int se3d_run_stage1_shard_pre(se3d_ctx *c, void *d_pack) {
  if (!c || !d_pack) return SE3D_ERR_ARG;
  if (c->staged_frames < 1) { c->err = "no frames staged"; return SE3D_ERR_STATE; }
  HIPCHK(c, hipSetDevice(c->device));
  Se3dDev D = c->D;
  D.d.B = c->staged_frames;
  se3d_launch_stage1(D, c->opt, c->pair_ab, c->stream, nullptr, 11, c->stream_hi, c->ev_hi, &c->plan, c->ev_chunk);   // association on every rank, triangulation of the own hypotheses
  se3d_launch_shard_pack_hyp(D, static_cast<se3d_person3d *>(d_pack), c->stream);
  HIPCHK(c, hipGetLastError());
  return SE3D_OK;
}
int se3d_run_stage1_shard_post(se3d_ctx *c, const void *d_gathered) {
  if (!c || !d_gathered) return SE3D_ERR_ARG;
  if (c->staged_frames < 1) return SE3D_ERR_STATE;
  HIPCHK(c, hipSetDevice(c->device));
  Se3dDev D = c->D;
  D.d.B = c->staged_frames;
  se3d_launch_shard_unpack_hyp(D, static_cast<const se3d_person3d *>(d_gathered), c->stream);
  se3d_launch_stage1(D, c->opt, c->pair_ab, c->stream, nullptr, 4, c->stream_hi, c->ev_hi, &c->plan, c->ev_chunk);   // post-processing + duplicate merge on every rank
  HIPCHK(c, hipGetLastError());
  c->ran = true;
  return SE3D_OK;
}
int se3d_run_stage2_shard_pre(se3d_ctx *c, int32_t n_seq, int32_t frame_slot0, const double *t, void *d_pack) {
  if (!c || !t || !d_pack || n_seq < 1 || frame_slot0 < 0) return SE3D_ERR_ARG;
  if (!c->ran || frame_slot0 + n_seq > c->staged_frames) { c->err = "se3d_run_stage2_shard_pre needs a completed stage 1 over >= n_seq frames"; return SE3D_ERR_STATE; }
  HIPCHK(c, hipSetDevice(c->device));
  Se3dStage2 Q = c->Q;
  double *ht = c->h_t + (size_t)c->t_head * Q.S;
  (void)hipEventSynchronize(c->t_done[c->t_head]);
  memcpy(ht, t, sizeof(double) * (size_t)n_seq);
  HIPCHK(c, hipMemcpyAsync(c->d_t, ht, sizeof(double) * (size_t)n_seq, hipMemcpyHostToDevice, c->stream));
  HIPCHK(c, hipEventRecord(c->t_done[c->t_head], c->stream));
  c->t_head = (c->t_head + 1) % se3d_ctx::kRing;
  Q.dets = c->D.outP + (size_t)frame_slot0 * Q.H; Q.n_det = c->D.out_count + frame_slot0; Q.fb_delay = c->d_fb + (size_t)frame_slot0 * Q.N; Q.t = c->d_t;
  c->Q_shard = Q;
  se3d_launch_stage2(Q, c->opt, 0, n_seq, c->stream, nullptr, 1);   // track association on every rank, LM of the own detections
  se3d_launch_shard_pack_lm(Q, n_seq, static_cast<Se3dShardRec *>(d_pack), c->stream);
  HIPCHK(c, hipGetLastError());
  return SE3D_OK;
}
int se3d_run_stage2_shard_post(se3d_ctx *c, int32_t n_seq, const void *d_gathered) {
  if (!c || !d_gathered || n_seq < 1) return SE3D_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  se3d_launch_shard_unpack_lm(c->Q_shard, n_seq, static_cast<const Se3dShardRec *>(d_gathered), c->stream);
  se3d_launch_stage2(c->Q_shard, c->opt, 0, n_seq, c->stream, nullptr, 2);   // merge-distance matrix, publish, lifecycle on every rank
  HIPCHK(c, hipGetLastError());
  c->ran2 = true;
  return SE3D_OK;
}

// ---- streaming ingest: ragged pinned H2D -> device expansion -> stage 1 -> walk -> device compaction -> D2H of the published persons ----
// ONE upload stream per device, shared by every context of the process (round 5).  tools/microbench/pcie_bw_multi.hip on the box: H2D alone
// 57.5 GB/s on any number of streams; H2D + D2H at once 56.5 + 45.5 GB/s with ONE stream per direction but 37 - 43 + 30 - 34 GB/s with two
// or more streams per direction (copies of both directions interleave on the same engines) — which is what eight contexts with a copy stream
// each produced (32 - 34.5 GB/s H2D whatever the number of contexts).  Uploads have no device-side dependency (a slot is reused only after its
// batch was collected), so a shared FIFO cannot block; it lives as long as the process.
// The download direction likewise: ONE shared stream per device.  A download waits for its batch's kernels; enqueued with a device-side wait it
// would block every other context's download behind it, so se3d_stream_collect waits for the batch on the HOST (compute_done), reads the number of
// published persons from the offsets the scan kernel mirrored into pinned memory, and only then enqueues the two copies.
// ---- the pair's health (round 6) ------------------------------------------------------------------------------------------------------------
// Round 5 saw one default bench.py run in about ten stream at 60 % of its rate for the WHOLE run: every step took what its uploads and its
// downloads take one after the other.  Round 6 found the mechanism with a probe on the pair (profiles/r6/README.md section 3): a healthy pair —
// both directions queued at once move 48 + 48 GB/s — turns into a serialised one (28 + 28 GB/s: both directions on one copy engine) after a burst of
// DOWNLOADS WITH THE UPLOAD STREAM IDLE, which is what the head of a streaming run produces when every feeding thread waits for its first batch,
// and stays serialised whatever runs afterwards; no copy pattern brings the overlap back, re-creating the two streams does, at once.  (Not
// reproducible with the same copies in a stand-alone program, tools/microbench/sdma_seq.hip: the trigger needs the process the contexts live in.)
// So the library watches the pair and repairs it:
//   * every download of at least 8 MB is bracketed by two timing events on the download stream (what lies between them is the copies alone; the
//     stream is a FIFO) — a healthy download runs at 37 - 56 GB/s with or without uploads next to it, one that shares its engine with the uploads
//     queues behind 200 MB uploads and comes out far below 30;
//   * PREVENTION (late round 6): the flip needs uploads alone FOLLOWED BY downloads alone (both, uploads alone, downloads alone, both: serialised every
//     time; the same with a small upload issued next to every download of the burst — 64 KB, 1 MB or 8 MB — never, in 12 of 12 probes:
//     profiles/r6/stream_companion.txt).  se3d_stream_collect therefore enqueues a 1 MB COMPANION upload (scratch to scratch, 0.5 % of a batch's upload
//     bytes) on the upload stream right before a batch's downloads: a download never starts with the upload stream idle;
//   * six downloads in a row below 30 GB/s: a NEW pair of streams replaces the old one (whose pending copies finish: hipStreamDestroy releases a
//     stream when its work is done; the events recorded on it stay valid).  At most 16 repairs per device and process.
// Every enqueue on the pair happens under g_copy_mu, so no thread can hold a handle that another thread destroys.
struct CopyPair {
  hipStream_t s[2] = {nullptr, nullptr};
  void *comp_h = nullptr, *comp_d = nullptr;   // companion upload (below): 1 MB of pinned host memory and of device memory
  int repairs = 0, slow_streak = 0;
  long downloads_timed = 0;
  double last_rate = 0.0;   // GB/s of the last timed download
};
static std::mutex g_copy_mu;
static CopyPair g_pair[64];
static constexpr double kPairSlowGBps = 30.0;
static constexpr int kPairSlowStreak = 6, kPairMaxRepairs = 16;
static constexpr size_t kPairTimedBytes = (size_t)8 << 20;
static constexpr size_t kPairCompanionBytes = (size_t)1 << 20;
// (caller holds g_copy_mu and has made `device` current)
static bool copy_pair_locked(int device, hipStream_t &up, hipStream_t &down) {
  if (device < 0 || device >= 64) return false;
  CopyPair &p = g_pair[device];
  for (int d = 0; d < 2; ++d)
    if (!p.s[d] && hipStreamCreateWithFlags(&p.s[d], hipStreamNonBlocking) != hipSuccess) p.s[d] = nullptr;
  up = p.s[0]; down = p.s[1];
  if (!p.comp_h && hipHostMalloc(&p.comp_h, kPairCompanionBytes, hipHostMallocDefault) == hipSuccess) std::memset(p.comp_h, 0, kPairCompanionBytes);
  if (!p.comp_d) (void)hipMalloc(&p.comp_d, kPairCompanionBytes);
  return up && down;
}
// (caller holds g_copy_mu) the companion upload of a download that is about to be enqueued; a failure is not an error of the caller's batch
static void copy_pair_companion_locked(int device) {
  CopyPair &p = g_pair[device];
  if (p.comp_h && p.comp_d && p.s[0]) (void)hipMemcpyAsync(p.comp_d, p.comp_h, kPairCompanionBytes, hipMemcpyHostToDevice, p.s[0]);
}
static void copy_pair_recreate_locked(int device) {
  CopyPair &p = g_pair[device];
  hipStream_t fresh[2] = {nullptr, nullptr};
  if (hipStreamCreateWithFlags(&fresh[0], hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&fresh[1], hipStreamNonBlocking) != hipSuccess) {
    if (fresh[0]) (void)hipStreamDestroy(fresh[0]);
    return;   // keep the old pair
  }
  for (int d = 0; d < 2; ++d) { if (p.s[d]) (void)hipStreamDestroy(p.s[d]); p.s[d] = fresh[d]; }
  ++p.repairs;
  p.slow_streak = 0;
}
// one timed download of `bytes` bytes took `ms` on the download stream (caller holds g_copy_mu)
static void copy_pair_note_download_locked(int device, size_t bytes, float ms) {
  CopyPair &p = g_pair[device];
  if (!(ms > 0.f)) return;
  p.last_rate = (double)bytes / ((double)ms * 1e6);
  ++p.downloads_timed;
  p.slow_streak = p.last_rate < kPairSlowGBps ? p.slow_streak + 1 : 0;
  if (p.slow_streak >= kPairSlowStreak && p.repairs < kPairMaxRepairs) copy_pair_recreate_locked(device);
}
// Four 32 MB uploads and four 32 MB downloads queued at once on the pair: GB/s of each direction over the wall time of the lot (~48 when the
// directions overlap, ~28 when they share an engine).  Both directions only: a burst of downloads alone is what breaks a healthy pair.
static bool probe_copy_pair(hipStream_t up, hipStream_t down, double *gbps_both) {
  const size_t n = (size_t)32 << 20;
  const int reps = 4;
  void *h0 = nullptr, *h1 = nullptr, *d0 = nullptr, *d1 = nullptr;
  bool ok = hipHostMalloc(&h0, n, hipHostMallocDefault) == hipSuccess && hipHostMalloc(&h1, n, hipHostMallocDefault) == hipSuccess &&
            hipMalloc(&d0, n) == hipSuccess && hipMalloc(&d1, n) == hipSuccess;
  auto run = [&](bool u, bool d) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < reps; ++k) {
      if (u) ok = ok && hipMemcpyAsync(d0, h0, n, hipMemcpyHostToDevice, up) == hipSuccess;
      if (d) ok = ok && hipMemcpyAsync(h1, d1, n, hipMemcpyDeviceToHost, down) == hipSuccess;
    }
    if (u) ok = ok && hipStreamSynchronize(up) == hipSuccess;
    if (d) ok = ok && hipStreamSynchronize(down) == hipSuccess;
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  };
  if (ok) {
#ifdef SE3D_PROBE_DIAG   // diagnostic builds only (tools/tmp, never shipped; -DSE3D_PROBE_DIAG=<bit mask>): 1 warm both, 2 uploads alone, 4 downloads alone,
    // 16 downloads with companion uploads of SE3D_PROBE_COMPANION_KB, 8 both (tools/r6.sh probe_bisect, tools/stream_companion_probe.sh)
#ifndef SE3D_PROBE_COMPANION_KB
#define SE3D_PROBE_COMPANION_KB 1024
#endif
    const int mode = SE3D_PROBE_DIAG;
    double tb = 1.0;
    if (mode & 1) (void)run(true, true);
    if (mode & 2) (void)run(true, false);
    if (mode & 4) (void)run(false, true);
    if (mode & 16) {   // downloads, each preceded by a small COMPANION upload: does that keep the burst from flipping the pair?
      const size_t small = (size_t)SE3D_PROBE_COMPANION_KB << 10;
      for (int k = 0; k < reps; ++k) {
        ok = ok && hipMemcpyAsync(d0, h0, small, hipMemcpyHostToDevice, up) == hipSuccess;
        ok = ok && hipMemcpyAsync(h1, d1, n, hipMemcpyDeviceToHost, down) == hipSuccess;
      }
      ok = ok && hipStreamSynchronize(up) == hipSuccess && hipStreamSynchronize(down) == hipSuccess;
    }
    if (mode & 8) tb = run(true, true);
#else
    (void)run(true, true);   // first touch of the buffers
    const double tb = run(true, true);
#endif
    *gbps_both = (double)reps * (double)n / 1e9 / tb;
  }
  if (h0) (void)hipHostFree(h0);
  if (h1) (void)hipHostFree(h1);
  if (d0) (void)hipFree(d0);
  if (d1) (void)hipFree(d1);
  return ok;
}
// Diagnostic (include/se3d.h): the state of the device's pair.
int se3d_copy_stream_probe(int32_t device, double *gbps_both, int32_t *repairs, double *last_download_gbps) {
  if (!gbps_both || device < 0 || device >= 64) return SE3D_ERR_ARG;
  if (hipSetDevice(device) != hipSuccess) return SE3D_ERR_DEVICE;
  std::lock_guard<std::mutex> lk(g_copy_mu);
  hipStream_t up = nullptr, down = nullptr;
  if (!copy_pair_locked(device, up, down)) return SE3D_ERR_DEVICE;
  if (!probe_copy_pair(up, down, gbps_both)) return SE3D_ERR_DEVICE;
  if (repairs) *repairs = g_pair[device].repairs;
  if (last_download_gbps) *last_download_gbps = g_pair[device].last_rate;
  return SE3D_OK;
}
static int stream_init(se3d_ctx *c) {
  if (c->stream_ready) return SE3D_OK;
  const size_t B = c->D.d.B, N = c->D.d.N, P = c->D.d.P, H = c->D.d.H;
  {
    std::lock_guard<std::mutex> lk(g_copy_mu);
    hipStream_t up = nullptr, down = nullptr;
    if (!copy_pair_locked(c->device, up, down)) { c->err = "se3d_stream_submit: no copy stream (at most 64 devices)"; return SE3D_ERR_DEVICE; }
  }
  for (auto &sl : c->ss) {
    HIPCHK(c, hipHostMalloc((void **)&sl.h_rag, sizeof(se3d_person2d) * B * N * P, hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc((void **)&sl.h_off, sizeof(int32_t) * B * (N + 1), hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc((void **)&sl.h_fused, sizeof(se3d_person3d) * B * H, hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc((void **)&sl.h_pred, sizeof(se3d_person3d) * B * H, hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc((void **)&sl.h_out_off, sizeof(int32_t) * (B + 1), hipHostMallocDefault));
    HIPCHK(c, hipHostMalloc((void **)&sl.h_fb, sizeof(float) * B * N, hipHostMallocDefault));
    HIPCHK(c, hipHostGetDevicePointer((void **)&sl.hd_fused, sl.h_fused, 0));
    HIPCHK(c, hipHostGetDevicePointer((void **)&sl.hd_pred, sl.h_pred, 0));
    HIPCHK(c, hipHostGetDevicePointer((void **)&sl.hd_out_off, sl.h_out_off, 0));
    if (!dev_alloc(c, &sl.d_fb, B * N)) return SE3D_ERR_DEVICE;
    if (!dev_alloc(c, &sl.d_rag, B * N * P) || !dev_alloc(c, &sl.d_off, B * (N + 1)) || !dev_alloc(c, &sl.d_fused, B * H) || !dev_alloc(c, &sl.d_pred, B * H) ||
        !dev_alloc(c, &sl.d_out_off, B + 1))
      return SE3D_ERR_DEVICE;
    HIPCHK(c, hipEventCreateWithFlags(&sl.in_done, hipEventDisableTiming));
    HIPCHK(c, hipEventCreateWithFlags(&sl.compute_done, hipEventDisableTiming));
    HIPCHK(c, hipEventCreate(&sl.off_done));     // timing events: they bracket the batch's download on the shared stream (copy_pair_note_download_locked)
    HIPCHK(c, hipEventCreate(&sl.d2h_start));
  }
  c->stream_ready = true;
  return SE3D_OK;
}

int se3d_stream_submit(se3d_ctx *c, int32_t n_frames, const se3d_frame_in *frames, int32_t n_seq, int32_t n_steps, const int32_t *frame_slot0, const double *t) {
  if (!c || !frames || !t || !frame_slot0 || n_frames < 1 || n_seq < 1 || n_steps < 1 || n_steps > SE3D_WALK_MAX_STEPS) return SE3D_ERR_ARG;
  const int B = c->D.d.B, N = c->D.d.N, P = c->D.d.P;
  if (n_frames > B) { c->err = "se3d_stream_submit: batch exceeds limits.max_frames"; return SE3D_ERR_CAPACITY; }
  // every frame slot of the batch must be visited by the walk: a slot it skips would publish the results of an EARLIER batch
  if ((size_t)n_seq * n_steps != (size_t)n_frames) { c->err = "se3d_stream_submit: n_seq * n_steps must equal n_frames"; return SE3D_ERR_ARG; }
  HIPCHK(c, hipSetDevice(c->device));
  int rc = stream_init(c);
  if (rc != SE3D_OK) return rc;
  se3d_ctx::StreamSlot &sl = c->ss[c->ss_head];
  if (sl.busy) { c->err = "se3d_stream_submit: two batches are already in flight (collect one first)"; return SE3D_ERR_STATE; }
  // ---- host: offsets, then the persons of every frame as ONE block into pinned memory (worker threads over frame ranges) ----
  int32_t total = 0;
  for (int f = 0; f < n_frames; ++f) {
    const se3d_frame_in &fr = frames[f];
    if (fr.n_cam != N || !fr.det_offset || (fr.det_offset[N] > 0 && !fr.persons)) { c->err = "bad frame"; return SE3D_ERR_ARG; }
    for (int cam = 0; cam <= N; ++cam) sl.h_off[(size_t)f * (N + 1) + cam] = total + fr.det_offset[cam];
    for (int cam = 0; cam < N; ++cam) {
      const int n = fr.det_offset[cam + 1] - fr.det_offset[cam];
      if (n < 0) { c->err = "det_offset not monotone"; return SE3D_ERR_ARG; }
      if (n > P) { c->err = "detections per camera exceed limits.max_det_per_cam"; return SE3D_ERR_CAPACITY; }
      sl.h_fb[(size_t)f * N + cam] = fr.fb_delay ? fr.fb_delay[cam] : -1.0f;
    }
    total += fr.det_offset[N];
  }
  {
    const int nthr = std::max(1, std::min<int>(8, std::min<int>((int)std::thread::hardware_concurrency(), n_frames / 256 + 1)));
    auto work = [&](int f0, int f1) {
      for (int f = f0; f < f1; ++f) {
        const int n = frames[f].det_offset[N];
        if (n) memcpy(sl.h_rag + sl.h_off[(size_t)f * (N + 1)], frames[f].persons, sizeof(se3d_person2d) * (size_t)n);
      }
    };
    std::vector<std::thread> th;
    for (int k = 1; k < nthr; ++k) th.emplace_back(work, (int)((int64_t)n_frames * k / nthr), (int)((int64_t)n_frames * (k + 1) / nthr));
    work(0, n_frames / nthr);
    for (auto &x : th) x.join();
  }
  // ---- H2D on the copy stream (no padding crosses PCIe), ordered after the previous use of this slot's device buffers ----
  {
    std::lock_guard<std::mutex> lk(g_copy_mu);   // (the pair may be replaced by another thread's repair: handles are read and used under the lock)
    hipStream_t up = nullptr, down = nullptr;
    if (!copy_pair_locked(c->device, up, down)) { c->err = "se3d_stream_submit: no copy stream"; return SE3D_ERR_DEVICE; }
    HIPCHK(c, hipMemcpyAsync(sl.d_rag, sl.h_rag, sizeof(se3d_person2d) * (size_t)total, hipMemcpyHostToDevice, up));
    HIPCHK(c, hipMemcpyAsync(sl.d_off, sl.h_off, sizeof(int32_t) * (size_t)n_frames * (N + 1), hipMemcpyHostToDevice, up));
    HIPCHK(c, hipMemcpyAsync(sl.d_fb, sl.h_fb, sizeof(float) * (size_t)n_frames * N, hipMemcpyHostToDevice, up));
    HIPCHK(c, hipEventRecord(sl.in_done, up));
  }
  // ---- compute stream: expand, stage 1, walk, compact ----
  HIPCHK(c, hipStreamWaitEvent(c->stream, sl.in_done, 0));
  Se3dDev D = c->D;
  D.d.B = n_frames;
  se3d_launch_stream_scatter(D, sl.d_rag, sl.d_off, c->stream);
  c->staged_frames = n_frames;
  se3d_launch_stage1(D, c->opt, c->pair_ab, c->stream, next_ev1(c), 15, c->stream_hi, c->ev_hi, &c->plan, c->ev_chunk);
  c->ran = true;
  c->d_fb_walk = sl.d_fb;
  rc = se3d_run_stage2_walk(c, n_seq, n_steps, frame_slot0, t);
  c->d_fb_walk = nullptr;
  if (rc != SE3D_OK) return rc;
  // Published persons back to the host.  The scan kernel mirrors the offsets into the slot's pinned buffer (a few KB of posted PCIe writes), so the
  // host knows the size without a copy + round trip.  Default: se3d_stream_collect downloads the persons on the device's shared download stream.
  // options.stream_out = 1: the compaction kernel WRITES the persons straight into the pinned host buffers (posted PCIe writes from the shader; the
  // kernel's end releases them at system scope): no copy engine at all — measured slower than the shared download stream (~28 GB/s of shader writes).
  sl.direct = c->opt.stream_out == 1;
  if (sl.direct) se3d_launch_stream_compact(c->Q.fused, c->Q.pred, c->Q.n_out, n_frames, c->Q.H, sl.d_out_off, sl.hd_out_off, sl.hd_fused, sl.hd_pred, c->stream);
  else se3d_launch_stream_compact(c->Q.fused, c->Q.pred, c->Q.n_out, n_frames, c->Q.H, sl.d_out_off, sl.hd_out_off, sl.d_fused, sl.d_pred, c->stream);
  HIPCHK(c, hipGetLastError());
  HIPCHK(c, hipEventRecord(sl.compute_done, c->stream));
  sl.n_frames = n_frames;
  sl.busy = true;
  c->ss_head ^= 1;
  return SE3D_OK;
}

int se3d_stream_collect(se3d_ctx *c, se3d_stream_result *out) {
  if (!c || !out) return SE3D_ERR_ARG;
  se3d_ctx::StreamSlot &sl = c->ss[c->ss_tail];
  if (!c->stream_ready || !sl.busy) { c->err = "se3d_stream_collect: no batch in flight"; return SE3D_ERR_STATE; }
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipEventSynchronize(sl.compute_done));   // the batch's kernels are done; the offsets are in pinned memory (mirror)
  const int total = sl.h_out_off[sl.n_frames];
  if (!sl.direct && total > 0) {   // (direct: the kernels wrote the pinned buffers themselves)
    const size_t bytes = 2 * sizeof(se3d_person3d) * (size_t)total;
    {
      std::lock_guard<std::mutex> lk(g_copy_mu);
      hipStream_t up = nullptr, down = nullptr;
      if (!copy_pair_locked(c->device, up, down)) { c->err = "se3d_stream_collect: no copy stream"; return SE3D_ERR_DEVICE; }
      copy_pair_companion_locked(c->device);   // the upload stream is never idle when a download starts (CopyPair)
      HIPCHK(c, hipEventRecord(sl.d2h_start, down));
      HIPCHK(c, hipMemcpyAsync(sl.h_fused, sl.d_fused, sizeof(se3d_person3d) * (size_t)total, hipMemcpyDeviceToHost, down));
      HIPCHK(c, hipMemcpyAsync(sl.h_pred, sl.d_pred, sizeof(se3d_person3d) * (size_t)total, hipMemcpyDeviceToHost, down));
      HIPCHK(c, hipEventRecord(sl.off_done, down));
    }
    HIPCHK(c, hipEventSynchronize(sl.off_done));   // (an event, not the stream: the stream also carries the other contexts' downloads)
    float ms = 0.f;
    if (bytes >= kPairTimedBytes && hipEventElapsedTime(&ms, sl.d2h_start, sl.off_done) == hipSuccess) {   // the pair's health: see CopyPair
      std::lock_guard<std::mutex> lk(g_copy_mu);
      copy_pair_note_download_locked(c->device, bytes, ms);
    }
  }
  out->n_frames = sl.n_frames;
  out->n_persons = total;
  out->person_offset = sl.h_out_off;
  out->fused = sl.h_fused;
  out->fused_pred = sl.h_pred;
  sl.busy = false;
  c->ss_tail ^= 1;
  return SE3D_OK;
}

int se3d_fetch_stage2(se3d_ctx *c, int32_t n_seq, se3d_fused_out *outs) {
  if (!c || !outs || n_seq < 1 || n_seq > c->Q.S) return SE3D_ERR_ARG;
  if (!c->ran2) { c->err = "fetch before run"; return SE3D_ERR_STATE; }
  HIPCHK(c, hipSetDevice(c->device));
  return fetch_stage2_range(c, 0, n_seq, outs);
}

int se3d_pack_stage2_device(se3d_ctx *c, int32_t n_seq, void *d_persons, void *d_counts) {
  if (!c || !d_persons || !d_counts || n_seq < 1 || n_seq > c->Q.S) return SE3D_ERR_ARG;
  if (!c->ran2) return SE3D_ERR_STATE;
  const Se3dStage2 &Q = c->Q;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipMemcpyAsync(d_persons, Q.fused, sizeof(se3d_person3d) * (size_t)n_seq * Q.H, hipMemcpyDeviceToDevice, c->stream));
  HIPCHK(c, hipMemcpyAsync(d_counts, Q.n_out, sizeof(int32_t) * (size_t)n_seq, hipMemcpyDeviceToDevice, c->stream));
  return SE3D_OK;  // asynchronous on the context stream (ordered after the se3d_run_stage2 that produced the data)
}

// ---- feedback reprojection (REPROJ:139-235) ----
int se3d_set_reprojection_cameras(se3d_ctx *c, const double *T, const double *P, const int32_t *wh) {
  if (!c || !T || !P || !wh) return SE3D_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  const int N = c->D.d.N;
  const int cap = std::max(c->Q.H, c->Q.T);
  if (!c->rp_ready) {
    double *cam = nullptr; int32_t *dwh = nullptr;
    if (!dev_alloc(c, &cam, (size_t)N * 16) || !dev_alloc(c, &dwh, (size_t)N * 2) || !dev_alloc(c, &c->d_rp_in, (size_t)cap) ||
        !dev_alloc(c, &c->R.dense, (size_t)N * cap) || !dev_alloc(c, &c->R.valid, (size_t)N * cap) || !dev_alloc(c, &c->R.slot, (size_t)N * cap) ||
        !dev_alloc(c, &c->R.det_offset, (size_t)N + 1) || !dev_alloc(c, &c->R.out, (size_t)N * cap) || !dev_alloc(c, &c->R.out_src, (size_t)N * cap))
      return SE3D_ERR_DEVICE;
    HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&c->h_rp_off), sizeof(int32_t) * (N + 1)));
    c->R.cam = cam; c->R.wh = dwh;
    c->R.N = N; c->R.T = cap; c->R.pose_method = c->params.pose_method;
  }
  std::vector<double> cam((size_t)N * 16);
  for (int i = 0; i < N; ++i) {
    for (int k = 0; k < 12; ++k) cam[(size_t)i * 16 + k] = T[(size_t)i * 16 + k];
    for (int k = 0; k < 4; ++k) cam[(size_t)i * 16 + 12 + k] = P[(size_t)i * 4 + k];
  }
  HIPCHK(c, hipStreamSynchronize(c->stream));
  HIPCHK(c, hipMemcpy(const_cast<double *>(c->R.cam), cam.data(), sizeof(double) * cam.size(), hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(const_cast<int32_t *>(c->R.wh), wh, sizeof(int32_t) * 2 * N, hipMemcpyHostToDevice));
  c->rp_ready = true;
  return SE3D_OK;
}

namespace {
int reproject_fetch(se3d_ctx *c, se3d_reproj_out *out) {
  const int N = c->R.N;
  HIPCHK(c, hipMemcpyAsync(c->h_rp_off, c->R.det_offset, sizeof(int32_t) * (N + 1), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::memcpy(out->det_offset, c->h_rp_off, sizeof(int32_t) * (N + 1));
  const int tot = c->h_rp_off[N];
  if (tot > out->capacity) { c->err = "se3d_reproject: output capacity too small"; return SE3D_ERR_CAPACITY; }
  if (tot > 0) {
    HIPCHK(c, hipMemcpy(out->persons, c->R.out, sizeof(se3d_person2d) * (size_t)tot, hipMemcpyDeviceToHost));
    if (out->person_index) HIPCHK(c, hipMemcpy(out->person_index, c->R.out_src, sizeof(int32_t) * (size_t)tot, hipMemcpyDeviceToHost));
  }
  return SE3D_OK;
}
}  // namespace

int se3d_reproject(se3d_ctx *c, int32_t n_persons, const se3d_person3d *persons, se3d_reproj_out *out) {
  if (!c || !out || !out->det_offset || (out->capacity > 0 && !out->persons) || n_persons < 0 || (n_persons > 0 && !persons)) return SE3D_ERR_ARG;
  if (!c->rp_ready) { c->err = "se3d_reproject: se3d_set_reprojection_cameras was not called"; return SE3D_ERR_STATE; }
  if (n_persons > c->R.T) { c->err = "se3d_reproject: more persons than limits.max_hyp / max_tracks"; return SE3D_ERR_CAPACITY; }
  HIPCHK(c, hipSetDevice(c->device));
  if (n_persons > 0) HIPCHK(c, hipMemcpyAsync(c->d_rp_in, persons, sizeof(se3d_person3d) * (size_t)n_persons, hipMemcpyHostToDevice, c->stream));
  Se3dReproj R = c->R;
  R.dense_form = c->opt.reproj_form;
  R.in = c->d_rp_in; R.n = n_persons; R.n_dev = nullptr;
  HIPCHK(c, se3d_launch_reproject(R, n_persons, c->stream));
  return reproject_fetch(c, out);
}

int se3d_reproject_resident(se3d_ctx *c, int32_t slot, se3d_reproj_out *out) {
  const int seq = slot;  // result slot: the sequence index after se3d_run_stage2 / se3d_refine, the FRAME slot after se3d_run_stage2_walk
  if (!c || seq < 0 || seq >= c->Q.S) return SE3D_ERR_ARG;
  if (!c->rp_ready) { c->err = "se3d_reproject_resident: se3d_set_reprojection_cameras was not called"; return SE3D_ERR_STATE; }
  if (!c->ran2) { c->err = "se3d_reproject_resident: no stage-2 result"; return SE3D_ERR_STATE; }
  HIPCHK(c, hipSetDevice(c->device));
  Se3dReproj R = c->R;
  R.dense_form = c->opt.reproj_form;
  R.in = c->Q.pred + (size_t)seq * c->Q.H; R.n = 0; R.n_dev = c->Q.n_out + seq;
  HIPCHK(c, se3d_launch_reproject(R, c->Q.H, c->stream));
  if (!out) return SE3D_OK;  // asynchronous: result stays on the device (timing / chaining)
  if (!out->det_offset || (out->capacity > 0 && !out->persons)) return SE3D_ERR_ARG;
  return reproject_fetch(c, out);
}

// All result slots slot0 .. slot0 + n_slots - 1 of the resident stage-2 results in ONE launch (the reprojection of a replayed batch: one
// list of predicted skeletons per frame slot -> N per-camera Person2D lists per slot).  Results stay on the device; se3d_fetch_reprojection
// copies one slot's lists out.
int se3d_reproject_resident_batch(se3d_ctx *c, int32_t slot0, int32_t n_slots) {
  if (!c || slot0 < 0 || n_slots < 1 || slot0 + n_slots > c->Q.S) return SE3D_ERR_ARG;
  if (!c->rp_ready) { c->err = "se3d_reproject_resident_batch: se3d_set_reprojection_cameras was not called"; return SE3D_ERR_STATE; }
  if (!c->ran2) { c->err = "se3d_reproject_resident_batch: no stage-2 result"; return SE3D_ERR_STATE; }
  HIPCHK(c, hipSetDevice(c->device));
  const size_t nt = (size_t)c->R.N * c->R.T;
  const bool want_lists = se3d_reproj_lists_lds(c->R.N, c->R.T, c->opt.reproj_form) > 0;
  if (n_slots > c->rpb_cap || want_lists != c->rpb_lists) {   // batch buffers: one block of every output array per slot
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // the batch buffers are OWNED by RB (never aliases of the single-list buffers of c->R): free what exists, member by member
    auto free_rb = [&]() {
      if (c->rpb_cap > 0 || c->RB.dense) {
        if (c->RB.dense) (void)hipFree(c->RB.dense);
        if (c->RB.valid) (void)hipFree(c->RB.valid);
        if (c->RB.slot) (void)hipFree(c->RB.slot);
        if (c->RB.det_offset) (void)hipFree(c->RB.det_offset);
        if (c->RB.out) (void)hipFree(c->RB.out);
        if (c->RB.out_src) (void)hipFree(c->RB.out_src);
      }
      c->RB.dense = nullptr; c->RB.valid = nullptr; c->RB.slot = nullptr; c->RB.det_offset = nullptr; c->RB.out = nullptr; c->RB.out_src = nullptr;
      c->rpb_cap = 0;
    };
    free_rb();
    c->RB = c->R;   // cameras, sizes, pose method ...
    c->RB.dense = nullptr; c->RB.valid = nullptr; c->RB.slot = nullptr; c->RB.det_offset = nullptr; c->RB.out = nullptr; c->RB.out_src = nullptr;   // ... but none of its buffers
    const size_t ni = want_lists ? 1 : nt * n_slots;   // the one-kernel form has no dense intermediate (1.3 GB per 16 384 slots at 16 x 12)
    if (hipMalloc(reinterpret_cast<void **>(&c->RB.dense), sizeof(se3d_person2d) * ni) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&c->RB.valid), ni) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&c->RB.slot), sizeof(int32_t) * ni) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&c->RB.det_offset), sizeof(int32_t) * (size_t)(c->R.N + 1) * n_slots) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&c->RB.out), sizeof(se3d_person2d) * nt * n_slots) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&c->RB.out_src), sizeof(int32_t) * nt * n_slots) != hipSuccess) {
      (void)hipGetLastError();
      free_rb();   // nothing half-allocated stays behind, nothing of c->R is touched
      c->err = "se3d_reproject_resident_batch: out of device memory";
      return SE3D_ERR_DEVICE;
    }
    c->rpb_cap = n_slots;
    c->rpb_lists = want_lists;
  }
  Se3dReproj R = c->RB;
  R.dense_form = c->opt.reproj_form;
  R.cam = c->R.cam; R.wh = c->R.wh; R.N = c->R.N; R.T = c->R.T; R.pose_method = c->R.pose_method;
  R.in = c->Q.pred + (size_t)slot0 * c->Q.H; R.in_stride = c->Q.H; R.n = 0; R.n_dev = c->Q.n_out + slot0;
  c->rpb_slot0 = slot0; c->rpb_n = n_slots;
  HIPCHK(c, se3d_launch_reproject(R, c->Q.H, c->stream, n_slots));
  return SE3D_OK;
}

int se3d_fetch_reprojection(se3d_ctx *c, int32_t slot, se3d_reproj_out *out) {
  if (!c || !out || !out->det_offset || (out->capacity > 0 && !out->persons)) return SE3D_ERR_ARG;
  if (c->rpb_n < 1 || slot < c->rpb_slot0 || slot >= c->rpb_slot0 + c->rpb_n) { c->err = "se3d_fetch_reprojection: slot not in the last batch"; return SE3D_ERR_STATE; }
  HIPCHK(c, hipSetDevice(c->device));
  const int N = c->R.N, b = slot - c->rpb_slot0;
  const size_t nt = (size_t)N * c->R.T;
  HIPCHK(c, hipMemcpyAsync(c->h_rp_off, c->RB.det_offset + (size_t)b * (N + 1), sizeof(int32_t) * (N + 1), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  std::memcpy(out->det_offset, c->h_rp_off, sizeof(int32_t) * (N + 1));
  const int tot = c->h_rp_off[N];
  if (tot > out->capacity) { c->err = "se3d_fetch_reprojection: output capacity too small"; return SE3D_ERR_CAPACITY; }
  if (tot > 0) {
    HIPCHK(c, hipMemcpy(out->persons, c->RB.out + b * nt, sizeof(se3d_person2d) * (size_t)tot, hipMemcpyDeviceToHost));
    if (out->person_index) HIPCHK(c, hipMemcpy(out->person_index, c->RB.out_src + b * nt, sizeof(int32_t) * (size_t)tot, hipMemcpyDeviceToHost));
  }
  return SE3D_OK;
}

extern "C" hipError_t se3d_read_fusion_constants(int32_t *indices22, double *vel_sigmas21);
int se3d_get_fusion_constants(se3d_ctx *c, int32_t *indices22, double *vel_sigmas21) {
  if (!c || !indices22 || !vel_sigmas21) return SE3D_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, se3d_read_fusion_constants(indices22, vel_sigmas21));
  return SE3D_OK;
}

// Diagnostic: the instruction-level square root / quotient of the pair-cost kernels against the IEEE expressions, on this device.
extern "C" hipError_t se3d_launch_selftest_exact_math(unsigned long long *d_out, unsigned long long n_div, hipStream_t st);
int se3d_selftest_exact_math(se3d_ctx *c, uint64_t n_div, uint64_t *sqrt_mismatches, uint64_t *div_mismatches) {
  if (!c || !sqrt_mismatches || !div_mismatches) return SE3D_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  unsigned long long *d = nullptr, h[2] = {0, 0};
  HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&d), sizeof(h)));
  hipError_t e = hipMemsetAsync(d, 0, sizeof(h), c->stream);
  if (e == hipSuccess) e = se3d_launch_selftest_exact_math(d, (unsigned long long)n_div, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  HIPCHK(c, e);
  *sqrt_mismatches = h[0]; *div_mismatches = h[1];
  return SE3D_OK;
}

// Diagnostic: the rows kernel's three-instruction quotient on the shared reciprocal against the IEEE quotient, on this device.
extern "C" hipError_t se3d_launch_selftest_shared_quotient(unsigned long long *d_out, unsigned long long n, hipStream_t st);
int se3d_selftest_shared_quotient(se3d_ctx *c, uint64_t n, uint64_t *mismatches, uint64_t *skipped) {
  if (!c || !mismatches || !skipped) return SE3D_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  unsigned long long *d = nullptr, h[2] = {0, 0};
  HIPCHK(c, hipMalloc(reinterpret_cast<void **>(&d), sizeof(h)));
  hipError_t e = hipMemsetAsync(d, 0, sizeof(h), c->stream);
  if (e == hipSuccess) e = se3d_launch_selftest_shared_quotient(d, (unsigned long long)n, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(h, d, sizeof(h), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d);
  HIPCHK(c, e);
  *mismatches = h[0]; *skipped = h[1];
  return SE3D_OK;
}

// Diagnostic: camera-column chunks of the pair table in this context (0: the whole table is materialised).
int se3d_get_pair_chunks(se3d_ctx *c) { return c ? c->D.d.pair_chunks : SE3D_ERR_ARG; }

// Diagnostic: the second pass's float32 DLT solve on caller systems (se3d_k_selftest_dlt32).
extern "C" hipError_t se3d_launch_selftest_dlt32(int n_sys, int n, const float *d_Rt, const float *d_pts, int weighted, float *d_ws, float *d_X, double *d_err, hipStream_t st);
int se3d_selftest_dlt32(se3d_ctx *c, int32_t n_sys, int32_t n_views, const float *Rt, const float *pts, int32_t weight_by_conf, float *X_out, double *err_out) {
  if (!c || n_sys < 1 || n_views < 2 || n_views > SE3D_MAX_CAMS || !Rt || !pts || !X_out || !err_out) return SE3D_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  const size_t nb = (size_t)(n_sys + 63) / 64;
  float *d_Rt = nullptr, *d_pts = nullptr, *d_ws = nullptr, *d_X = nullptr;
  double *d_err = nullptr;
  hipError_t e = hipMalloc(reinterpret_cast<void **>(&d_Rt), sizeof(float) * 12 * n_views * n_sys);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_pts), sizeof(float) * 3 * n_views * n_sys);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_ws), sizeof(float) * nb * 64 * 8 * n_views);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_X), sizeof(float) * 3 * n_sys);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&d_err), sizeof(double) * n_sys);
  if (e == hipSuccess) e = hipMemcpyAsync(d_Rt, Rt, sizeof(float) * 12 * n_views * n_sys, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(d_pts, pts, sizeof(float) * 3 * n_views * n_sys, hipMemcpyHostToDevice, c->stream);
  if (e == hipSuccess) e = se3d_launch_selftest_dlt32(n_sys, n_views, d_Rt, d_pts, weight_by_conf, d_ws, d_X, d_err, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(X_out, d_X, sizeof(float) * 3 * n_sys, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(err_out, d_err, sizeof(double) * n_sys, hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(d_Rt); (void)hipFree(d_pts); (void)hipFree(d_ws); (void)hipFree(d_X); (void)hipFree(d_err);
  HIPCHK(c, e);
  return SE3D_OK;
}

// Diagnostic: the joints the last triangulation pass of this context handed to the float32-exact second pass.
int se3d_get_exact_list(se3d_ctx *c, uint32_t *entries, int32_t capacity, int32_t *n_entries, int64_t *total) {
  if (!c || !n_entries || capacity < 0 || (capacity > 0 && !entries)) return SE3D_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  uint32_t h[4] = {0, 0, 0, 0};
  HIPCHK(c, hipMemcpyAsync(h, c->D.tri_cnt, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *n_entries = (int32_t)h[0];
  if (total) *total = (int64_t)h[1];
  const size_t n = std::min<size_t>((size_t)h[0], (size_t)capacity);
  if (n) {
    HIPCHK(c, hipMemcpyAsync(entries, c->D.tri_list, sizeof(uint32_t) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
  }
  return SE3D_OK;
}

// Diagnostic: how often the per-camera sweep of this context (large rigs: se3d_k_sweep_assign) met an ambiguous step (S3D:628-634).
int se3d_get_sweep_stats(se3d_ctx *c, uint64_t *steps, uint64_t *hungarian_steps) {
  if (!c || !steps || !hungarian_steps) return SE3D_ERR_ARG;
  HIPCHK(c, hipSetDevice(c->device));
  uint32_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  HIPCHK(c, hipMemcpyAsync(h, c->D.tri_cnt, sizeof(h), hipMemcpyDeviceToHost, c->stream));
  HIPCHK(c, hipStreamSynchronize(c->stream));
  *steps = h[4]; *hungarian_steps = h[5];
  return SE3D_OK;
}

int se3d_profile_enable(se3d_ctx *c, int32_t on) {
  if (!c) return SE3D_ERR_ARG;
  collect_profile(c);
  c->prof = on != 0;
  return SE3D_OK;
}
int se3d_profile_reset(se3d_ctx *c) {
  if (!c) return SE3D_ERR_ARG;
  collect_profile(c);
  for (int i = 0; i < PS_COUNT; ++i) { c->prof_ms[i] = 0; c->prof_n[i] = 0; }
  return SE3D_OK;
}
int se3d_profile_get(se3d_ctx *c, const char *name, double *ms_total, int64_t *launches) {
  if (!c || !name) return SE3D_ERR_ARG;
  collect_profile(c);
  for (int i = 0; i < PS_COUNT; ++i)
    if (strcmp(name, kProfNames[i]) == 0) {
      if (ms_total) *ms_total = c->prof_ms[i];
      if (launches) *launches = c->prof_n[i];
      return SE3D_OK;
    }
  return SE3D_ERR_ARG;
}

}  // extern "C"
