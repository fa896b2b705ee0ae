// se3d_stage1.hip — CDNA4 (gfx950) kernels for the skeleton_3d hot path: normalisation, pairwise epipolar cost table,
// greedy cross-view association (cost gather + assignment with device Munkres), per-joint DLT triangulation with
// outlier rejection and unscented-transform covariance, per-person post-processing and duplicate merge.
// Reference rows: N S3D:312-333 · E S3D:335-390 · G S3D:565-674 · M HUN:60-397 · T S3D:440-465 ·
// P S3D:425-438 · O S3D:748-844 · U S3D:471-523 · X S3D:849-973 · D S3D:392-423,984-996.
//
// Compile with -ffp-contract=off: the float32 geometry that decides association indices must round exactly like the
// reference's SSE2 build (no FMA); fp64 kernels that want FMA call __builtin_fma explicitly.
#include <type_traits>
#include "se3d_device.h"
#include "se3d_munkres.h"
#include "se3d_exact32.h"

#include <cstdlib>
#include <cstring>

// Diagnostic build only (make diag): cycle stamps of se3d_k_sweep_assign (never in the product .so)
#ifdef SE3D_LM_STAMPS
__device__ unsigned long long g_s1_stamps[16];
#define S1_STAMP(idx)                                                                  \
  {                                                                                    \
    const unsigned long long now_ = __builtin_readcyclecounter();                      \
    if (threadIdx.x == 0) atomicAdd(&g_s1_stamps[idx], now_ - s1_prev_);               \
    s1_prev_ = __builtin_readcyclecounter();                                           \
  }
#define S1_STAMP_INIT unsigned long long s1_prev_ = __builtin_readcyclecounter();
#define S1_COUNT(idx) if (threadIdx.x == 0) atomicAdd(&g_s1_stamps[idx], 1ull);
extern "C" void se3d_diag_mk_stamps(unsigned long long *out16, int reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_mk_stamps), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_mk_stamps), z, sizeof(z)); }
}
extern "C" void se3d_diag_s1_stamps(unsigned long long *out16, int reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_s1_stamps), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_s1_stamps), z, sizeof(z)); }
}
// ... and of se3d_k_pairs_rows: cycles of wavefront 0 (stages operands) and of wavefront 3 (does not) of every 16th workgroup, by phase part
__device__ unsigned long long g_rb_stamps[2][16];
#define RB_STAMP_INIT unsigned long long rb_prev_ = __builtin_readcyclecounter(); const bool rb_on_ = (tid & 63) == 0 && ((tid >> 6) == 0 || (tid >> 6) == 3) && ((blockIdx.y + 3 * blockIdx.z) & 15) == 0;
#define RB_STAMP(idx) { const unsigned long long now_ = __builtin_readcyclecounter(); if (rb_on_) atomicAdd(&g_rb_stamps[(tid >> 6) == 3][idx], now_ - rb_prev_); rb_prev_ = __builtin_readcyclecounter(); }
#define RB_COUNT(idx) if (rb_on_) atomicAdd(&g_rb_stamps[(tid >> 6) == 3][idx], 1ull);
extern "C" void se3d_diag_rb_stamps(unsigned long long *out32, int reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_rb_stamps), sizeof(unsigned long long) * 32);
  if (reset) { unsigned long long z[32] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rb_stamps), z, sizeof(z)); }
}
#else
#define S1_STAMP(idx)
#define S1_STAMP_INIT
#define S1_COUNT(idx)
#define RB_STAMP_INIT
#define RB_STAMP(idx)
#define RB_COUNT(idx)
#endif

#define TRI_THRESH 0.30f
#define REPROJ_MAX 0.050
#define MAX_COSTS 1e6

// =============================================================================================================
// row N — normalize_keypoints (S3D:312-333) + ">= 9 valid keypoints" filter (S3D:579, 599) with order-preserving
// compaction of the surviving detections.  One workgroup per (camera, frame).
// =============================================================================================================
// The per-frame workspace reset covers assoc / nviews / nhyp / flags only: the triangulation kernels write every field of the hypothesis
// slots they own (round 1 zero-filled hypP here: 166 MB of stores per 8192-frame launch at 16 cameras x 6 persons).
// write_kp = 0: masks / slots only (config 2, where no pair kernel reads the normalised points).
__global__ __launch_bounds__(256) void se3d_k_normalize(Se3dDev D, int write_kp) {
  const int c = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
  const int N = D.d.N, P = D.d.P;
  __shared__ int cnt[SE3D_MUNKRES_MAX];
  __shared__ int slot[SE3D_MUNKRES_MAX];
  __shared__ unsigned mgt[SE3D_MUNKRES_MAX], mge[SE3D_MUNKRES_MAX];
  __shared__ int s_nvalid;
  const int n = min(D.nraw[f * N + c], P);
  for (int r = tid; r < P; r += blockDim.x) { cnt[r] = 0; mgt[r] = 0; mge[r] = 0; }
  if (!(write_kp & 2)) {  // per-frame workspace reset: the N workgroups of a frame share the work (write_kp & 2: keep it — se3d_get_pair_costs after a pass)
    const int H = D.d.H;
    for (int h = tid; h < H; h += blockDim.x) D.assoc[((size_t)f * H + h) * N + c] = -1;
    if (c == 0) {
      for (int h = tid; h < H; h += blockDim.x) D.nviews[f * H + h] = 0;
      if (tid == 0) { D.nhyp[f] = 0; D.flags[f] = 0u; }
    }
  }
  __syncthreads();
  const se3d_person2d *raw = D.raw + ((size_t)(f * N + c)) * P;
  for (int idx = tid; idx < n * SE3D_NKP; idx += blockDim.x) {
    const int det = idx / SE3D_NKP, j = idx - det * SE3D_NKP;
    const float sc = raw[det].keypoints[j].score;
    if (sc >= TRI_THRESH) {
      atomicAdd(&cnt[det], 1);
      atomicOr(&mge[det], 1u << j);
      if (sc > TRI_THRESH) atomicOr(&mgt[det], 1u << j);
    }
  }
  __syncthreads();
  if (tid == 0) {
    int k = 0;
    for (int det = 0; det < n; ++det) slot[det] = (cnt[det] > SE3D_NKP / 2) ? k++ : -1;
    s_nvalid = k;
  }
  __syncthreads();
  const float fx = D.K[c * 4 + 0], fy = D.K[c * 4 + 1], cx = D.K[c * 4 + 2], cy = D.K[c * 4 + 3];
  const size_t base = ((size_t)(f * N + c)) * P;
  for (int idx = tid; (write_kp & 1) && idx < n * SE3D_NKP; idx += blockDim.x) {
    const int det = idx / SE3D_NKP, j = idx - det * SE3D_NKP;
    const int k = slot[det];
    if (k < 0) continue;
    const se3d_keypoint2d kp = raw[det].keypoints[j];
    float4 o = make_float4(0.f, 0.f, -1.f, 0.f);
    if (kp.score >= TRI_THRESH) o = make_float4((kp.x - cx) / fx, (kp.y - cy) / fy, kp.score, 0.f);
    D.kp[(base + k) * SE3D_NKP + j] = o;
  }
  for (int det = tid; det < n; det += blockDim.x) {
    const int k = slot[det];
    if (k >= 0) {
      D.pscore[base + k] = raw[det].score;
      D.orig[base + k] = det;
      D.vmask_gt[base + k] = mgt[det];
      D.vmask_ge[base + k] = mge[det];
    }
  }
  if (tid == 0) D.nvalid[f * N + c] = s_nvalid;
}

// ---- correctly rounded float32 square root, reciprocal and quotient without the special-case scaffolding --------------------------------------
// sqrtf(x) and a / b compile to the IEEE sequences of the AMDGPU back end: v_sqrt_f32 + a one-ulp correction from two FMA residuals, and
// v_rcp_f32 + six FMA / multiply refinements — wrapped in operand scaling (v_div_scale_f32, x * 2^32), v_div_fmas / v_div_fixup and class
// tests for denormal, huge, zero and infinite operands: 16 and ~14 instructions.  The pair cost needs two of each per (entry, joint).
// Rounds 2 - 4 used the arithmetic cores of those sequences (9 + 8 instructions and two transcendental ones per distance n / sqrtf(s)).
// Round 5: ONE v_rsq_f32 and 9 more instructions per distance, each step decided by enumeration or by a theorem:
//   * y = rsq(s); s0 = s y; h = y / 2; root = fma(fma(-s0, s0, s), h, s0) — one Newton correction of a 1-ulp estimate — returns sqrtf(s) on
//     EVERY float in [2^-60, 2^60] (1 006 632 961 values; the same with v_sqrt_f32 and a reciprocal as the correction's factor fails on 60);
//   * rcp = fma(fma(-root, y, 1), y, y) — one Newton step on the same seed — is RN(1 / root) for every one of those radicands except the 120
//     whose root has an all-ones significand (1 / root lies 2^-24 ulp from a rounding boundary there);
//   * the quotient is the rows kernel's se3d_div_shared on that reciprocal: q0 = RN(n rcp), r = fma(-root, q0, n), q = fma(r, rcp, q0) — the
//     correctly rounded n / root by the one-correction theorem stated above se3d_k_pairs_rows (root not all ones; here root in [2^-30, 2^30]
//     and n zero or in [2^-60, 2^60], so q0 and the residual's quantum 2^-46 n stay normal).
// (The back end's refinement seeded with y instead of v_rcp_f32 — two instructions fewer than rounds 2 - 4, no reciprocal — matched a / b on
// 2^34 random operand pairs and FAILS on 14 278 numerators against all-ones roots: tools/microbench/div_rsq_seed.hip.  Random pairs do not
// decide such a sequence.)  se3d_selftest_exact_math runs the two enumerations on the device (tests/test_gpu_stage1.py),
// tools/microbench/sqrt_candidates.hip holds the candidates that were compared.  Inside the operand range in which the scaffolding is the identity
// (se3d_pair_ops_in_range) and for roots that are not all ones these return the bits of the IEEE expressions; otherwise the callers take the
// IEEE expressions themselves.
__device__ __forceinline__ float se3d_sqrt_rcp_core(float x, float &rcp) {   // returns sqrtf(x); rcp = RN(1 / sqrtf(x)) unless the root is all ones
  const float y = __builtin_amdgcn_rsqf(x);
  const float s0 = x * y, h = 0.5f * y;
  const float b = __builtin_fmaf(__builtin_fmaf(-s0, s0, x), h, s0);
  rcp = __builtin_fmaf(__builtin_fmaf(-b, y, 1.f), y, y);
  return b;
}
__device__ __forceinline__ bool se3d_all_ones(float b) { return (__float_as_uint(b) & 0x7fffffu) == 0x7fffffu; }
#define SE3D_EXACT_LO 0x21800000u   // 2^-60
#define SE3D_EXACT_HI 0x5d800000u   // 2^60
// s1, s2: arguments of the square roots (their roots, in [2^-30, 2^30], are the divisors); n1, n2 >= 0: the numerators.  Non-negative floats
// order like their bit patterns; n - 1 wraps zero to the top, so a zero numerator passes and anything in (0, 2^-60) does not.
__device__ __forceinline__ bool se3d_pair_ops_in_range(float s1, float s2, float n1, float n2) {
  const unsigned a = __float_as_uint(s1), b = __float_as_uint(s2), c = __float_as_uint(n1), d = __float_as_uint(n2);
  const unsigned hi = max(max(a, b), max(c, d)), lo_s = min(a, b), lo_n = min(c - 1u, d - 1u);
  return hi <= SE3D_EXACT_HI && lo_s >= SE3D_EXACT_LO && lo_n >= SE3D_EXACT_LO - 1u;
}
__device__ __forceinline__ float se3d_div_shared(float a, float b, float y);   // a / b from y = RN(1 / b): defined with the rows kernel
// d1 + d2 of one joint of one (det_a, det_b) pair: the two point-to-epipolar-line distances (S3D:358-361)
__device__ __forceinline__ float se3d_pair_joint_dist(float n1, float s1, float n2, float s2) {
  if (se3d_pair_ops_in_range(s1, s2, n1, n2)) {
    float y1, y2;
    const float b1 = se3d_sqrt_rcp_core(s1, y1), b2 = se3d_sqrt_rcp_core(s2, y2);
    if (!(se3d_all_ones(b1) || se3d_all_ones(b2))) return se3d_div_shared(n1, b1, y1) + se3d_div_shared(n2, b2, y2);
  }
  return n1 / sqrtf(s1) + n2 / sqrtf(s2);
}

// =============================================================================================================
// row E (inner) — symmetric epipolar distance of one (view-a detection, view-b detection) pair, exactly the float
// expression tree of calcCost (S3D:353-365) with Eigen's fixed-size evaluation order a0 + (a1 + a2).
// =============================================================================================================
__device__ __forceinline__ double se3d_pair_cost(const float *__restrict__ F, const float4 *__restrict__ ka, const float4 *__restrict__ kb,
                                                 unsigned m) {
  double cost = 0.0;
  int n = 0;
  const float f0 = F[0], f1 = F[1], f2 = F[2], f3 = F[3], f4 = F[4], f5 = F[5], f6 = F[6], f7 = F[7], f8 = F[8];
#pragma unroll
  for (int k = 0; k < SE3D_NKP; ++k) {
    if ((m >> k) & 1u) {
      const float4 A = ka[k], Bq = kb[k];
      const float p1x = A.x, p1y = A.y, p2x = Bq.x, p2y = Bq.y;
      const float l1x = f0 * p1x + (f1 * p1y + f2);
      const float l1y = f3 * p1x + (f4 * p1y + f5);
      const float l1z = f6 * p1x + (f7 * p1y + f8);
      const float l2x = f0 * p2x + (f3 * p2y + f6);
      const float l2y = f1 * p2x + (f4 * p2y + f7);
      const float l2z = f2 * p2x + (f5 * p2y + f8);
      cost += (double)se3d_pair_joint_dist(fabsf(p2x * l1x + (p2y * l1y + l1z)), l1x * l1x + l1y * l1y,
                                           fabsf(p1x * l2x + (p1y * l2y + l2z)), l2x * l2x + l2y * l2y);
      ++n;
    }
  }
  return n > 0 ? cost / n : -1.0;
}

// flat variant: one thread per (frame, pair, det_a, det_b); used for small P
__global__ __launch_bounds__(256) void se3d_k_pairs_flat(Se3dDev D, const int2 *__restrict__ pair_ab) {
  const int N = D.d.N, P = D.d.P, np = D.d.npairs;
  const size_t total = (size_t)D.d.B * np * P * P;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int q = (int)(i % P);
    const int p = (int)((i / P) % P);
    const int pr = (int)((i / ((size_t)P * P)) % np);
    const int f = (int)(i / ((size_t)P * P * np));
    const int2 ab = pair_ab[pr];
    if (p >= D.nvalid[f * N + ab.x] || q >= D.nvalid[f * N + ab.y]) continue;
    const size_t ia = ((size_t)(f * N + ab.x)) * P + p, ib = ((size_t)(f * N + ab.y)) * P + q;
    const unsigned m = D.vmask_gt[ia] & D.vmask_gt[ib];
    D.pair[i] = se3d_pair_cost(D.F + 9 * (size_t)pr, D.kp + ia * SE3D_NKP, D.kp + ib * SE3D_NKP, m);
  }
}

// Small rigs (N * P <= 256): normalisation and the packed pair-cost table in ONE workgroup per frame.  The normalised points of the
// frame's detections live only in LDS (float2 x, y per (camera, slot, joint): 35 KB at most) — the kp / kcov copies of round 1 (two
// float4 per joint: 428 MB of stores and 642 MB of loads per 8192-frame launch at 16 x 6) are gone; the triangulation kernels read the
// caller's raw Person2D payload themselves.  Same float expressions as se3d_k_normalize / se3d_pair_cost -> identical table.
// acc / n for a small integer n (the joint count 1..17 of a pair entry, S3D:365; the view counts of the sweep, S3D:616-626), correctly
// rounded (= the IEEE quotient) in three fp64 operations instead of the ~35-instruction IEEE division sequence: with y = RN(1 / n) = (1 + d) / n,
// q0 = RN(acc y), r = acc - n q0 (exact in an FMA: q0 is within 1.5 ulp of acc / n) the exact value of q0 + r y is acc / n + d (acc / n - q0), i.e.
// within 1.5 x 2^-53 ulp of the quotient, while acc / n keeps a distance of at least ulp / (2 n) from every rounding boundary (acc - n m is a
// non-zero multiple of ulp / 2 for a midpoint m): the final rounding is the quotient's for every n < 2^50 (Markstein 1990; Thm. 8.5 in Muller
// et al., "Handbook of Floating-Point Arithmetic"); tests/test_markstein64.py checks it against IEEE division on the CPU.
// An infinite (or NaN) accumulator — a degenerate epipolar line, norm 0 — keeps IEEE division's result: inf - inf inside the residual would
// turn acc = inf into NaN where acc / n, the reference and the flat kernel give inf.
__device__ __forceinline__ double se3d_div_count(double acc, int n, double inv_n) {
  const double q0 = acc * inv_n;
  const double r = __fma_rn(-(double)n, q0, acc);
  return __builtin_isfinite(q0) ? __fma_rn(r, inv_n, q0) : q0;
}

#define NP_MAX_DETS 256
// inv[n] = RN(1 / n), n = 1 .. 17 (LDS): the entry's cost / n by se3d_div_count
__device__ __forceinline__ double se3d_pair_cost_lds(const float *__restrict__ F, const float2 *__restrict__ ka, const float2 *__restrict__ kb, unsigned m, const double *inv) {
  double cost = 0.0;
  int n = 0;
  const float f0 = F[0], f1 = F[1], f2 = F[2], f3 = F[3], f4 = F[4], f5 = F[5], f6 = F[6], f7 = F[7], f8 = F[8];
#pragma unroll
  for (int k = 0; k < SE3D_NKP; ++k) {
    if ((m >> k) & 1u) {
      const float2 A = ka[k], Bq = kb[k];
      const float p1x = A.x, p1y = A.y, p2x = Bq.x, p2y = Bq.y;
      const float l1x = f0 * p1x + (f1 * p1y + f2);
      const float l1y = f3 * p1x + (f4 * p1y + f5);
      const float l1z = f6 * p1x + (f7 * p1y + f8);
      const float l2x = f0 * p2x + (f3 * p2y + f6);
      const float l2y = f1 * p2x + (f4 * p2y + f7);
      const float l2z = f2 * p2x + (f5 * p2y + f8);
      cost += (double)se3d_pair_joint_dist(fabsf(p2x * l1x + (p2y * l1y + l1z)), l1x * l1x + l1y * l1y,
                                           fabsf(p1x * l2x + (p1y * l2y + l2z)), l2x * l2x + l2y * l2y);
      ++n;
    }
  }
  return n > 0 ? se3d_div_count(cost, n, inv[n]) : -1.0;
}

// Rows N of a frame for the small-rig kernels below: per-frame workspace reset, valid-joint masks, order-preserving slot compaction, the
// normalised points of the kept detections into LDS.  256 threads; ends with the points visible to the whole workgroup.
// nt: threads of the workgroup (>= N * P); s_psc (optional): the detection scores by filtered slot, for a sweep that follows in the same kernel.
__device__ __forceinline__ void se3d_np_prologue(const Se3dDev &D, int f, int tid, float2 *skp, unsigned *s_mgt, int *s_slot, int *s_nv, int nt = 256, float *s_psc = nullptr) {
  const int N = D.d.N, P = D.d.P, H = D.d.H;
  // ---- per-frame workspace reset (assoc, nviews, nhyp, flags) ----
  for (int i = tid; i < H * N; i += nt) D.assoc[(size_t)f * H * N + i] = -1;
  for (int h = tid; h < H; h += nt) D.nviews[f * H + h] = 0;
  if (tid == 0) { D.nhyp[f] = 0; D.flags[f] = 0u; }
  // ---- row N: valid-joint counts and masks, one thread per (camera, raw detection) ----
  const int c = tid / P, det = tid - c * P;
  const bool mine = tid < N * P && det < min(D.nraw[f * N + (c < N ? c : 0)], P);
  unsigned mgt = 0u, mge = 0u;
  const se3d_person2d *rp = D.raw + ((size_t)(f * N + (c < N ? c : 0))) * P + det;
  if (mine) {
#pragma unroll
    for (int j = 0; j < SE3D_NKP; ++j) {
      const float sc = rp->keypoints[j].score;
      if (sc >= TRI_THRESH) { mge |= 1u << j; if (sc > TRI_THRESH) mgt |= 1u << j; }
    }
  }
  const bool kept = mine && __popc(mge) > SE3D_NKP / 2;  // S3D:579, 599
  if (tid < N * P) s_slot[tid] = kept ? 1 : 0;
  __syncthreads();
  int slot = -1;
  if (kept) {  // order-preserving compaction within the camera
    slot = 0;
    for (int d2 = 0; d2 < det; ++d2) slot += s_slot[c * P + d2];
  }
  int nv = 0;
  if (tid < N) for (int d2 = 0; d2 < P; ++d2) nv += s_slot[tid * P + d2];
  __syncthreads();
  if (tid < N * P) s_slot[tid] = slot;
  if (tid < N) { s_nv[tid] = nv; D.nvalid[f * N + tid] = nv; }
  if (kept) {
    const size_t o = ((size_t)(f * N + c)) * P + slot;
    D.pscore[o] = rp->score;
    D.orig[o] = det;
    D.vmask_gt[o] = mgt;
    D.vmask_ge[o] = mge;
    s_mgt[c * P + slot] = mgt;
    if (s_psc) s_psc[c * P + slot] = rp->score;
  }
  __syncthreads();
  // ---- normalised points of the kept detections into LDS ----
  for (int idx = tid; idx < N * P * SE3D_NKP; idx += nt) {
    const int cd = idx / SE3D_NKP, j = idx - cd * SE3D_NKP;
    const int cc = cd / P, dd = cd - cc * P;
    if (dd >= min(D.nraw[f * N + cc], P)) continue;
    const int k = s_slot[cd];
    if (k < 0) continue;
    const se3d_keypoint2d *kp = &D.raw[((size_t)(f * N + cc)) * P + dd].keypoints[j];
    const float sc = kp->score;
    float2 o = make_float2(0.f, 0.f);
    if (sc >= TRI_THRESH) {
      const float fx = D.K[cc * 4 + 0], fy = D.K[cc * 4 + 1], cx = D.K[cc * 4 + 2], cy = D.K[cc * 4 + 3];
      o = make_float2((kp->x - cx) / fx, (kp->y - cy) / fy);
    }
    skp[(size_t)(cc * P + k) * SE3D_NKP + j] = o;
  }
  __syncthreads();
}

__global__ __launch_bounds__(256) void se3d_k_normalize_pairs(Se3dDev D, const int2 *__restrict__ pair_ab) {
  extern __shared__ __align__(16) unsigned char np_smem[];
  const int N = D.d.N, P = D.d.P, np = D.d.npairs;
  const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float2 *skp = reinterpret_cast<float2 *>(np_smem);                       // [N * P][17] filtered slot order
  unsigned *s_mgt = reinterpret_cast<unsigned *>(skp + (size_t)N * P * SE3D_NKP);  // [N * P] by filtered slot
  int *s_slot = reinterpret_cast<int *>(s_mgt + N * P);                    // [N * P] by raw detection index: filtered slot or -1
  int *s_nv = s_slot + N * P;                                              // [N]
  int *s_cum = s_nv + N;                                                   // [np + 1]
  __shared__ int s_part[4];
  __shared__ double s_inv[SE3D_NKP + 1];   // RN(1 / n) for the joint count of an entry (visible after the prologue's barriers)
  if (tid <= SE3D_NKP) s_inv[tid] = tid ? 1.0 / (double)tid : 0.0;
  se3d_np_prologue(D, f, tid, skp, s_mgt, s_slot, s_nv);
  // ---- exclusive prefix of n_a * n_b over the camera pairs ----
  int carry = 0;
  for (int base = 0; base < np; base += 256) {
    const int pr = base + tid;
    int v = 0;
    if (pr < np) { const int2 ab = pair_ab[pr]; v = s_nv[ab.x] * s_nv[ab.y]; }
    int inc = v;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) s_part[wave] = inc;
    __syncthreads();
    int off = carry;
    for (int w = 0; w < wave; ++w) off += s_part[w];
    if (pr < np) s_cum[pr] = off + inc - v;
    carry += s_part[0] + s_part[1] + s_part[2] + s_part[3];
    __syncthreads();
  }
  if (tid == 0) s_cum[np] = carry;
  __syncthreads();
  // ---- row E: the valid (det_a, det_b) entries of all camera pairs, densely over the threads ----
  const int total = s_cum[np];
  double *const tabf = D.pair + (size_t)f * np * P * P;   // the frame's table: offsets inside it fit 24-bit multiplies (np <= 2048, N * P <= 256)
  for (int e = tid; e < total; e += 256) {
    int lo = 0, hi = np;  // largest pr with s_cum[pr] <= e
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (s_cum[mid] <= e) lo = mid; else hi = mid; }
    const int pr = lo;
    const int2 ab = pair_ab[pr];
    const int nb = s_nv[ab.y];
    const int r = e - s_cum[pr];
    const int p = r / nb, q = r - __mul24(p, nb);
    const int ia = __mul24(ab.x, P) + p, ib = __mul24(ab.y, P) + q;
    const unsigned m = s_mgt[ia] & s_mgt[ib];
    tabf[__mul24(__mul24(pr, P) + p, P) + q] = se3d_pair_cost_lds(D.F + __mul24(9, pr), skp + __mul24(ia, SE3D_NKP), skp + __mul24(ib, SE3D_NKP), m, s_inv);
  }
}

// Range guard of the shared-reciprocal divide (se3d_div_shared, rows kernel): every operand of a numerator / denominator is 0 or within
// [2^-30, 2^30].  Then a non-zero numerator is >= 2^-84 and a quotient >= 2^-115, so with float32 denormals preserved every FMA residual is
// exact (no underflow) and the correction returns the correctly rounded quotient; zero numerators give exactly 0.
// The test on the five floats of an operand at once: |v| as an integer orders like |v|, and |v| - 1 wraps zero to the top (v_min3 / v_max3).
__device__ __forceinline__ bool se3d_all_in_safe_range(float a, float b, float c, float d, float e) {
  const unsigned ua = __float_as_uint(a) & 0x7fffffffu, ub = __float_as_uint(b) & 0x7fffffffu, uc = __float_as_uint(c) & 0x7fffffffu,
                 ud = __float_as_uint(d) & 0x7fffffffu, ue = __float_as_uint(e) & 0x7fffffffu;
  const unsigned hi = max(max(max(ua, ub), uc), max(ud, ue));
  const unsigned lo = min(min(min(ua - 1u, ub - 1u), uc - 1u), min(ud - 1u, ue - 1u));
  return hi <= ((158u << 23) - 1u) && lo >= (97u << 23) - 1u;
}
// y = RN(1/b) for divisors where Markstein's theorem applies (normal range, significand not all ones), else 0
__device__ __forceinline__ float se3d_safe_rcp(float b) {
  const unsigned u = __float_as_uint(b);
  const unsigned ex = (u >> 23) & 0xffu;
  const bool ok = (ex >= 97u && ex <= 157u) && ((u & 0x7fffffu) != 0x7fffffu);
  return ok ? 1.0f / b : 0.0f;
}

// =============================================================================================================
// row E, large P ("rows"): one workgroup per (row tile of det_a, camera pair, frame), looping over 32-column tiles
// of det_b.  A wavefront carries TWO det_a rows (lanes 0-31 / 32-63) x 32 det_b columns:
//   * the det_b side (epipolar line F^T p2, its norm and shared reciprocal, the point, the validity weight: 8 floats per joint)
//     stays in REGISTERS for a group of 6 joints while the wavefront walks its rows;
//   * the det_a side is read from LDS with two (half-wave uniform) addresses per instruction instead of the 16 distinct ds_read_b128
//     per joint of the 4x4-register-tile kernel;
//   * 124 VGPRs, no scratch, 38 KB of LDS: four workgroups (16 wavefronts) per CU instead of two to three.
// Phases (column tile, joint group 6 / 6 / 5) are software-pipelined: the operands of phase k + 1 are written to the other half of a
// double-buffered LDS tile before phase k computes, the global loads of phase k + 2 are in flight meanwhile, the table entries of a
// finished column tile are stored one phase late, and the one barrier per phase orders LDS only.  Measured at 256 cam x 128 p per 16
// frames (profiles/r2): 4x4 tiles 85.8 ms; rows with 9 / 8 joint groups and __syncthreads() 67.0 ms (of which, by doubling the item loop:
// 39 ms items at the microbenchmarked issue floor, 28 ms everything else); the round-2 form 59.6 ms.
// Same per-element float expressions in the same order (FAST: shared-reciprocal divide under the block-uniform range guard,
// else IEEE divide) -> the table is bit-identical to every other pair kernel.
//
// Round 5 (profiles/r5/README.md section 1; per launch of 64 frames x 2040 pairs: 16.0 - 16.8 ms in round 4):
//   * the two quotients of an item take THREE instructions each instead of five (se3d_div_shared): with y = RN(1 / b) staged per operand,
//     q0 = RN(a y), r = fma(-b, q0, a), q = fma(r, y, q0) is the correctly rounded a / b.  Why ONE correction suffices (rounds 2 - 4 spent two,
//     following the textbook sequence "1.5 ulp -> faithful -> correct"): write x = a / b in [f, f + u), u = ulp, and b y = 1 + eta with
//     |eta| <= 2^-24 / sig(y).  Then |a y - x| = x |eta| < u, so q0 is at most 1.5 u from x, and the value the last FMA rounds is
//     x + (r / b) eta + (rounding of r, when q0 is not faithful and r needs 25 bits): off by less than 4 x 2^-24 u.  (A) x farther than that from
//     the midpoint f + u / 2: the rounding is x's.  (B) x within 4 x 2^-24 u of the midpoint: a q0 outside {f, f + u} would need
//     x |eta| >= u (1 - 4 x 2^-24), i.e. sig(q) / (2 sig(y)) >= 1 - 2^-22, i.e. sig(y) <= 1 + 2^-23 with |eta| at its bound — b = 2^24 - k
//     in units of its ulp gives |eta| 2^24 = 1 - k (k + 1) 2^-24 for odd k (tiny for even k), and the inequality leaves k = 1: the all-ones
//     significand, which the staging guard excludes (se3d_safe_rcp) — so q0 is FAITHFUL there, r is exact and Markstein's theorem (1990;
//     Muller et al., Handbook of Floating-Point Arithmetic, ch. 4: a faithful quotient corrected with the correctly rounded reciprocal) gives
//     the correctly rounded quotient.  Every operand is 0 or inside [2^-30, 2^30] (staging guard) and float32 denormals are preserved
//     (.amdhsa_float_denorm_mode_32 3), so no residual loses a bit to underflow.  Checked against the IEEE quotient: on the device on 2^33
//     operand pairs (se3d_selftest_shared_quotient, tests/test_gpu_stage1.py); on the CPU with exact float32 arithmetic on all 2^23
//     numerators of a binade for divisors at the edges of the guard, on 7 M quotients constructed to lie 2^-25 .. 2^-22 ulp from a midpoint and
//     on random operands (tests/test_shared_quotient_cpu.py).  18 instead of 22 vector instructions per item: 16.8 -> 15.2 ms;
//   * the row tiles of a camera are BALANCED (ceil(n_a / 48) tiles of equal size, the rows of a tile dealt evenly to the four wavefronts): the
//     bench scenes hold 65 - 110 detections per camera (mean 90, not 128), and 40 % of the cameras used to need a third tile of <= 14 rows whose
//     workgroup staged the whole det_b side for one busy wavefront;
//   * an operand's sqrtf and 1 / nrm are the instruction-level cores of the IEEE sequences inside their range (se3d_sqrt_rcp_core):
//     staging is a quarter of the kernel's vector instructions (the det_b side is staged once per row tile).
// What did NOT pay (measured, same section): one row x 64 columns per wavefront with the validity as scalar branch + lane mask (19 instead of 22
// instructions per item, but 64-column tiles compute 128 columns for every camera: 20.3 ms), and one 512-thread workgroup per camera pair with
// all rows in a single tile (det_b staged once; 75 KB of LDS: one workgroup per CU, 24.7 ms).  The kernel is bound by vector issue AND the LDS
// array together (8 LDS cycles per wave-item against 36 vector-issue cycles on each of four SIMDs).
// =============================================================================================================
#define RB_ROWS 48
#define RB_COLS 32
#define RB_NT 256                       // threads per workgroup
// Lanes per det_a row.  32 (default, rounds 2 - 6): a wavefront-instruction carries TWO rows x 32 columns, one det_b column per lane, and every item
// pays one det_a operand read (two ds_read_b128 with two half-wave-uniform addresses).  16 (round 6, -DRB_RL=16, kept as a compile-time form):
// FOUR rows x 16 lanes, each lane carrying TWO det_b columns (ql, ql + 16), so that one operand read serves two items per lane — the lever
// round 5's review named ("the LDS pipe, not the VALU, is the wall").  Built, bit-identical, and SLOWER in every joint grouping that fits
// (profiles/r6/ab/: 3 joints per phase 15.25 ms, 4: 14.98, 5 at three wavefronts per SIMD: 16.58, 6 at three: 15.58, against 14.02 - 14.08 for this
// layout on the same box).  The counters say why (profiles/r6/ab/pmc_*_rl32.json, *_rl16j3.json): the vector ALU is busy the whole time in BOTH
// forms (SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x cycles) = 1.19 / 1.08; 7.87 G / 8.52 G VALU wave-instructions per launch at ~3.4 cycles each — the
// item's v_cvt_f64_f32 and v_add_f64 hold the port 4.3 cycles, profiles/r2/microbench_valu_rates.txt), LDS instructions fall only from 0.97 G to
// 0.79 G (the det_b operands are fetched per lane and phase: twice as many with two columns) while VALU instructions rise 8 % and SALU 52 %
// (twice the phases).  The 5.7 ms that round 5's `nolds` ablation attributed to the LDS reads is read-to-use LATENCY inside a VALU-saturated
// schedule, not LDS bandwidth: the kernel is bound by vector issue, and fewer instructions per item is the only lever left.
#ifndef RB_RL
#define RB_RL 32
#endif
#define RB_CPL (RB_COLS / RB_RL)          // det_b columns per lane
#define RB_RPI (64 / RB_RL)               // det_a rows per wavefront-instruction
#define RB_WSTEPS (RB_ROWS / 4 / RB_RPI)  // row steps a wavefront can carry (RB_CPL accumulators each)
// LDS rows are padded by one operand (32 B = 8 banks): the staging threads of a wavefront write the SAME detection slot of consecutive
// joints, and an unpadded joint stride (48 or 32 operands) is a multiple of 128 B — every lane on the same four banks (17-way / 6-way
// conflicts, 29 % of the LDS busy cycles in profiles/r2/pmc_lds_256cam128p_rows.json before the padding)
#define RB_ROWS_P (RB_ROWS + 1)
#define RB_COLS_P (RB_COLS + 1)
#ifndef RB_JH
#define RB_JH (RB_RL == 16 ? 3 : 6)
#endif
#define RB_JLAST (SE3D_NKP % RB_JH ? SE3D_NKP % RB_JH : RB_JH)   // joints of the last phase
struct __align__(16) RowOp { float lx, ly, lz, nrm, x, y, rcp, z; };   // z: the point's homogeneous coordinate, 1 (usable) or 0

// a / b, correctly rounded, from y = RN(1 / b) (see above): 3 instructions
__device__ __forceinline__ float se3d_div_shared(float a, float b, float y) {
  const float q0 = a * y;
  const float r = __builtin_fmaf(-b, q0, a);
  return __builtin_fmaf(r, y, q0);
}

// Operand of one (detection, joint): epipolar line, |(lx, ly)|, its reciprocal, the point (x, y, z = 1; all zero when the joint is unusable).  sqrtf and 1 / nrm are the
// IEEE results: inside the range in which the instruction-level cores are the IEEE sequences (se3d_sqrt_rcp_core, checked by
// se3d_selftest_exact_math) the cores, else the library calls.
__device__ __forceinline__ RowOp se3d_rowop(float lx, float ly, float lz, float px, float py, bool valid, bool &unsafe) {
  RowOp o;
  o.lx = lx; o.ly = ly; o.lz = lz;
  // (an unusable operand — zero line, norm forced to 1 below — must not drag its wavefront through the IEEE branch: it computes on 1.0)
  const float ss = valid ? lx * lx + ly * ly : 1.f;
  const unsigned us = __float_as_uint(ss);
  if (us >= SE3D_EXACT_LO && us <= SE3D_EXACT_HI) {   // then nrm lies in [2^-30, 2^30]: se3d_safe_rcp's exponent test holds
    float y;
    o.nrm = se3d_sqrt_rcp_core(ss, y);
    o.rcp = se3d_all_ones(o.nrm) ? 0.f : y;
  } else {
    o.nrm = sqrtf(ss);
    o.rcp = se3d_safe_rcp(o.nrm);
  }
  o.x = px; o.y = py;
  o.z = valid ? 1.f : 0.f;
  // Range guard of the shared-reciprocal quotient: an operand outside it is marked by rcp = 0 (a usable operand's reciprocal is never 0); the
  // wavefronts that meet one take the IEEE-divide form for that phase.  Both forms give the correctly rounded quotients, hence the same bits, so
  // the choice need not be uniform over the workgroup (rounds 2 - 4 kept block-wide flags in LDS: one more LDS round trip per phase).
  const bool bad = valid && !(o.rcp != 0.f && se3d_all_in_safe_range(lx, ly, lz, px, py));
  unsafe |= bad;
  if (bad) o.rcp = 0.f;
  // An unusable (detection, joint) is all zeros — line, point AND the point's homogeneous coordinate z — with a unit norm.  The item forms the
  // numerators as |p.x l.x + (p.y l.y + p.z l.z)|: with p.z = 1 the FMA p.z l.z + t rounds exactly like the reference's addition l.z + t, and with
  // either side unusable BOTH numerators are exactly 0 (zero point x finite line, finite point x zero line), so the item is d1 + d2 — no validity
  // weights (rounds 2 - 5a spent a multiply and an FMA on d1 * w_b + d2 * w_a: 18 -> 17 vector instructions per item).  The IEEE-divide path
  // (se3d_rows_compute_slow) selects by the validity masks and never divides by this norm for a masked joint's contribution.
  if (!valid) { o.lx = 0.f; o.ly = 0.f; o.lz = 0.f; o.x = 0.f; o.y = 0.f; o.nrm = 1.f; o.rcp = 1.f; }
  return o;
}

template <int JN>
__device__ __forceinline__ void se3d_rows_compute(const RowOp (*A)[RB_ROWS_P], const RowOp (&b)[RB_CPL][RB_JH], int jbase, int row0, int nrows_left,
                                                  double (&acc)[RB_WSTEPS][RB_CPL]) {
  // Straight-line code over (row step, joint).  The det_a operand of the NEXT item is fetched from LDS while the current one is
  // computed (two RowOp in flight = 16 registers).  The memory-clobbering empty asm after every item is what keeps it that way: without
  // it the det_a operands — which do not change from one column tile to the next — are lifted out of the column loop by loop-invariant
  // code motion, and the independent LDS reads of the block are all issued first (102 reads = 816 registers, 2 KB of scratch per lane).
  asm volatile("" ::: "memory");
  RowOp a = A[jbase][row0];
#pragma unroll
  for (int r = 0; r < RB_WSTEPS; ++r) {
    // wave-uniform (scalar) skip of the row steps beyond the wavefront's last row.  NOT a `break`: the early exit made the compiler copy the
    // other accumulators to the exit block's registers before every row step (five 64-bit moves per six items).
    if (RB_RPI * r < nrows_left) {
    const int row = row0 + RB_RPI * r;
#pragma unroll
    for (int jj = 0; jj < JN; ++jj) {
      const bool last_j = jj + 1 == JN, last_r = r + 1 == RB_WSTEPS;
      const RowOp an = A[last_j ? jbase : jbase + jj + 1][last_j ? (last_r ? row0 : row + RB_RPI) : row];
#pragma unroll
      for (int c = 0; c < RB_CPL; ++c) {   // one det_a operand, RB_CPL det_b columns
        const float n1 = fabsf(b[c][jj].x * a.lx + __builtin_fmaf(b[c][jj].z, a.lz, b[c][jj].y * a.ly));   // z = 1: RN(a.lz + t), the reference's addition
        const float n2 = fabsf(a.x * b[c][jj].lx + __builtin_fmaf(a.z, b[c][jj].lz, a.y * b[c][jj].ly));
        const float d1 = se3d_div_shared(n1, a.nrm, a.rcp), d2 = se3d_div_shared(n2, b[c][jj].nrm, b[c][jj].rcp);
        const float sacc = d1 + d2;   // exactly 0 unless both sides have the joint (se3d_rowop)
        acc[r][c] += (double)sacc;
      }
      // the item is complete (accumulated) before the fence; the next item's LDS read comes after it
      if (RB_CPL == 2) asm volatile("" : "+v"(acc[r][0]), "+v"(acc[r][RB_CPL - 1]) : : "memory");
      else asm volatile("" : "+v"(acc[r][0]) : : "memory");
      a = an;
    }
    }
  }
}
// IEEE-divide form for blocks whose operands fail the range guard of the shared-reciprocal divide (practically never): compact loops, both
// operands from LDS.
__device__ __forceinline__ void se3d_rows_compute_slow(const RowOp (*A)[RB_ROWS_P], const RowOp (*Bc)[RB_COLS_P], int ql, int jbase, int jn, int row0,
                                                       const unsigned (&m2)[RB_CPL], const unsigned *M1, double (&acc)[RB_WSTEPS][RB_CPL]) {
#pragma unroll
  for (int r = 0; r < RB_WSTEPS; ++r) {
    const int row = row0 + RB_RPI * r;
#pragma unroll
    for (int c = 0; c < RB_CPL; ++c) {
      const unsigned mm = M1[row] & m2[c];
#pragma unroll 1
      for (int jj = 0; jj < jn; ++jj) {
        const RowOp a = A[jbase + jj][row], bq = Bc[jj][ql + RB_RL * c];
        const float n1 = fabsf(bq.x * a.lx + (bq.y * a.ly + a.lz));
        const float n2 = fabsf(a.x * bq.lx + (a.y * bq.ly + bq.lz));
        const float sacc = n1 / a.nrm + n2 / bq.nrm;
        acc[r][c] += ((mm >> (jbase + jj)) & 1u) ? (double)sacc : 0.0;
      }
    }
  }
}

// Workgroup barrier that orders LDS traffic only (s_waitcnt lgkmcnt(0); s_barrier).
__device__ __forceinline__ void se3d_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifndef RB_WPE
#define RB_WPE 4
#endif
// list0 / slot0 / fbase: the workgroups of grid row y work on pair list0 + y of `pair_ab` and write slot slot0 + y of the frame's pair slots;
// frame = fbase + blockIdx.z.  Whole table: list0 = slot0 = 0 over all pairs; column chunks (Se3dPairPlan): one launch per chunk.
// gridDim.x = row tiles a camera can need: ceil(P / RB_ROWS).
__global__ __launch_bounds__(RB_NT, RB_WPE) void se3d_k_pairs_rows(Se3dDev D, const int2 *__restrict__ pair_ab, int list0, int slot0, int fbase) {
  const int N = D.d.N, P = D.d.P, np = D.d.pair_stride;
  const int f = fbase + blockIdx.z;
  const int2 ab = pair_ab[list0 + blockIdx.y];
  const int pr = slot0 + blockIdx.y;
  const int na_all = D.nvalid[f * N + ab.x], nb = D.nvalid[f * N + ab.y];
  const int ntiles = (na_all + RB_ROWS - 1) / RB_ROWS;               // balanced row tiles of the camera's detections
  if ((int)blockIdx.x >= ntiles || nb == 0) return;
  const int tile_rows = ((na_all + ntiles - 1) / ntiles + RB_RPI - 1) & ~(RB_RPI - 1);   // a wavefront-instruction carries RB_RPI rows
  const int p0 = blockIdx.x * tile_rows;
  if (p0 >= na_all) return;
  const int na = min(na_all, p0 + tile_rows);                        // rows p0 .. na - 1 are this tile's
  // The tile's rows dealt to the four wavefronts (even shares, a wavefront carries row pairs).  NOT evenly: wavefronts 0 - 2 also stage the det_b
  // operands of the next phase (RB_COLS x RB_JH = 192 threads) — 17 % of the kernel's time by ablation, as long as ~3.7 row pairs of items by the
  // cycle stamps (tools/rb_stamps.py) — while wavefront 3 waited at the barrier for them (35 % of its cycles).  So wavefront 3 takes its full
  // 12 rows first and the staging wavefronts share the rest: nrw rows each, the smallest even number with 3 nrw + 12 >= rows
  // (14.31 -> 14.15 ms per launch at 256cam128p; with seven accumulators and 14 rows for wavefront 3: 14.23, 128 VGPRs).
  const int nt = na - p0;
  const int nrw = min(RB_ROWS / 4, max(RB_RPI, ((max(0, nt - RB_ROWS / 4) + 2) / 3 + RB_RPI - 1) & ~(RB_RPI - 1)));   // rows of each of wavefronts 0 - 2
  const int nr3 = min(RB_ROWS / 4, RB_ROWS - 3 * nrw);                              // rows wavefront 3 can take (>= nt - 3 nrw by construction)
  __shared__ RowOp A[SE3D_NKP][RB_ROWS_P];  // det_a side, all joints: staged once per workgroup (26 KB)
  __shared__ RowOp Bc[2][RB_JH][RB_COLS_P];  // det_b side of a (column tile, joint group) phase, double-buffered (2 x 6 KB)
  __shared__ unsigned M1[RB_ROWS], M2[2][RB_COLS];
  __shared__ int s_unsafe_a;   // range-guard flag of the det_a side (the det_b side: rcp = 0 in the operand, tested from registers)
  __shared__ double INV[SE3D_NKP + 1];     // RN(1 / n) for the joint count of an entry
  const int tid = threadIdx.x, lane = tid & 63, sub = lane / RB_RL, ql = lane % RB_RL;   // sub: the lane's row inside a wavefront-instruction
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // scalar: everything derived from it is a uniform branch, not an exec mask
  const float *F = D.F + 9 * (size_t)se3d_pair_index(ab.x, ab.y, N);   // F is stored in the reference's pair order (get_fundamental_idx)
  const float f0 = F[0], f1 = F[1], f2 = F[2], f3 = F[3], f4 = F[4], f5 = F[5], f6 = F[6], f7 = F[7], f8 = F[8];
  const size_t baseA = ((size_t)(f * N + ab.x)) * P, baseB = ((size_t)(f * N + ab.y)) * P;
  RB_STAMP_INIT
  RB_COUNT(15)
  if (tid == 0) s_unsafe_a = 0;
  if (tid <= SE3D_NKP) INV[tid] = tid ? 1.0 / (double)tid : 0.0;
  constexpr int NA_ROUNDS = (RB_ROWS * SE3D_NKP + RB_NT - 1) / RB_NT;
  float4 a4[NA_ROUNDS];   // all global loads of the det_a side first (in flight across the barrier): one memory latency, not five in a row
#pragma unroll
  for (int u = 0; u < NA_ROUNDS; ++u) {
    const int i = tid + 256 * u, d = i / SE3D_NKP, k = i - d * SE3D_NKP, p = p0 + d;
    a4[u] = make_float4(0.f, 0.f, -1.f, 0.f);
    if (i < RB_ROWS * SE3D_NKP && p < na) a4[u] = D.kp[(baseA + p) * SE3D_NKP + k];
  }
  for (int d = tid; d < RB_ROWS; d += 256) M1[d] = (p0 + d < na) ? D.vmask_gt[baseA + p0 + d] : 0u;
  __syncthreads();
  RB_STAMP(0)   // entry: F, counts, det_a global loads + first barrier
  {
    bool unsafe = false;
#pragma unroll
    for (int u = 0; u < NA_ROUNDS; ++u) {
      const int i = tid + 256 * u, d = i / SE3D_NKP, k = i - d * SE3D_NKP, p = p0 + d;
      if (i < RB_ROWS * SE3D_NKP)
        A[k][d] = se3d_rowop(f0 * a4[u].x + (f1 * a4[u].y + f2), f3 * a4[u].x + (f4 * a4[u].y + f5), f6 * a4[u].x + (f7 * a4[u].y + f8),
                             a4[u].x, a4[u].y, p < na && ((M1[d] >> k) & 1u), unsafe);
    }
    if (unsafe) s_unsafe_a = 1;
  }
  const int row0 = wave * nrw + sub;                         // this lane's first row inside the tile; row steps r: row0 + RB_RPI r
  const int nshare = wave == 3 ? nr3 : nrw;                  // rows this wavefront owns (wave-uniform)
  const int nrows_left = min(nshare, nt - wave * nrw);       // ... and those of them that exist
  double *out = D.pair + ((size_t)f * np + pr) * P * P;
  double acc[RB_WSTEPS][RB_CPL];
  unsigned m2[RB_CPL];
#pragma unroll
  for (int c = 0; c < RB_CPL; ++c) m2[c] = 0u;
  // One (detection, joint) operand of the det_b side per thread and phase (RB_COLS x RB_JH <= 256).  Its global loads are issued a whole
  // phase ahead (before the barrier and the compute of the current phase), and the entries of a finished column tile are stored one
  // phase late, after those loads have been consumed: loads and stores retire through one in-order counter (vmcnt), so a load issued
  // behind the epilogue's stores would also wait for their round trip to L2.
  static_assert(RB_COLS * RB_JH <= 256, "one staging item per thread");
  float4 pre4 = make_float4(0.f, 0.f, -1.f, 0.f);
  unsigned premq = 0u;
  // the thread's (column, joint) of a phase with RB_JH / RB_JLAST joints: constant over the phases
  const int sd6 = tid / RB_JH, sj6 = tid - sd6 * RB_JH, sd5 = tid / RB_JLAST, sj5 = tid - sd5 * RB_JLAST;
  const float4 *kpB = D.kp + baseB * SE3D_NKP;
  const uint32_t *vmB = D.vmask_gt + baseB;
  auto prefetch = [&](int q0, int jbase, int jn) {
    const int d = jn == RB_JH ? sd6 : sd5, jj = jn == RB_JH ? sj6 : sj5, q = q0 + d;
    pre4 = make_float4(0.f, 0.f, -1.f, 0.f);
    premq = 0u;
    if (tid < RB_COLS * jn && q < nb) { pre4 = kpB[q * SE3D_NKP + jbase + jj]; premq = vmB[q]; }
  };
  auto flush = [&](int q0_done) {   // the LDS reads of the entries first (batches of independent reads), then the arithmetic and the stores
    int nn[RB_WSTEPS][RB_CPL];
    double iv[RB_WSTEPS][RB_CPL];
#pragma unroll
    for (int r = 0; r < RB_WSTEPS; ++r)
#pragma unroll
      for (int c = 0; c < RB_CPL; ++c) nn[r][c] = __popc(M1[row0 + RB_RPI * r] & m2[c]);
#pragma unroll
    for (int r = 0; r < RB_WSTEPS; ++r)
#pragma unroll
      for (int c = 0; c < RB_CPL; ++c) iv[r][c] = INV[nn[r][c]];
    double *o = out + (size_t)(p0 + row0) * P + q0_done + ql;
#pragma unroll
    for (int r = 0; r < RB_WSTEPS; ++r) {
      const int p = p0 + row0 + RB_RPI * r;
#pragma unroll
      for (int c = 0; c < RB_CPL; ++c)
        if (RB_RPI * r < nshare && p < na && q0_done + ql + RB_RL * c < nb)   // (row steps beyond the wavefront's share belong to the next wavefront)
          o[(size_t)(RB_RPI * r) * P + RB_RL * c] = nn[r][c] > 0 ? se3d_div_count(acc[r][c], nn[r][c], iv[r][c]) : -1.0;
    }
  };
  // Phase ph = (column tile ph / NPH, joint group ph % NPH).  While a phase computes from registers, the operands of the NEXT phase are
  // already in the other half of Bc (written before the compute from the prefetched registers) and the loads of the phase after that are
  // in flight: ONE barrier per phase.
  constexpr int NPH = (SE3D_NKP + RB_JH - 1) / RB_JH;
  const int nph = NPH * ((nb + RB_COLS - 1) / RB_COLS);
  auto jn_of = [](int ph_) { return min(RB_JH, SE3D_NKP - (ph_ % NPH) * RB_JH); };
  auto stage = [&](int ph_) {   // pre4 / premq (operand of phase ph_) -> Bc[ph_ & 1], M2[(ph_ / NPH) & 1], flag[ph_ % 3]
    const int jn = jn_of(ph_), jbase = (ph_ % NPH) * RB_JH, q0 = (ph_ / NPH) * RB_COLS;
    const int d = jn == RB_JH ? sd6 : sd5, jj = jn == RB_JH ? sj6 : sj5, k = jbase + jj, q = q0 + d;
    bool unsafe = false;
    if (tid < RB_COLS * jn) {
      Bc[ph_ & 1][jj][d] = se3d_rowop(f0 * pre4.x + (f3 * pre4.y + f6), f1 * pre4.x + (f4 * pre4.y + f7), f2 * pre4.x + (f5 * pre4.y + f8),
                                      pre4.x, pre4.y, q < nb && ((premq >> k) & 1u), unsafe);
      if (jj == 0 && jbase == 0) M2[(ph_ / NPH) & 1][d] = premq;
    }
  };
  auto prefetch_ph = [&](int ph_) {
    if (ph_ >= nph) return;
    const int jbase = (ph_ % NPH) * RB_JH, q0 = (ph_ / NPH) * RB_COLS;
    if (jn_of(ph_) == RB_JH) prefetch(q0, jbase, RB_JH); else prefetch(q0, jbase, RB_JLAST);
  };
  RB_STAMP(1)   // det_a operands (RowOp of the tile's rows) into LDS
  prefetch_ph(0);
  stage(0);
  prefetch_ph(1);
  se3d_lds_barrier();   // A, M1, the flag and phase 0 of Bc are visible
  const int unsafe_a = __builtin_amdgcn_readfirstlane(s_unsafe_a);   // read ONCE: the det_a side is staged once per workgroup
  RB_STAMP(2)   // first det_b phase: load + stage + barrier
  for (int ph = 0; ph < nph; ++ph) {
    const int jn = jn_of(ph), jbase = (ph % NPH) * RB_JH, q0 = (ph / NPH) * RB_COLS;
    RB_COUNT(14)
    RB_STAMP(3)   // flag read
    if (ph + 1 < nph) stage(ph + 1);   // consumes the prefetched registers before any store of this phase is issued
    RB_STAMP(4)   // stage the next phase's det_b operands
    if (jbase == 0) {
      if (ph > 0) flush(q0 - RB_COLS);
#pragma unroll
      for (int r = 0; r < RB_WSTEPS; ++r)
#pragma unroll
        for (int c = 0; c < RB_CPL; ++c) acc[r][c] = 0.0;
#pragma unroll
      for (int c = 0; c < RB_CPL; ++c) m2[c] = M2[(ph / NPH) & 1][ql + RB_RL * c];
    }
    RB_STAMP(5)   // flush of the finished column tile + accumulator reset
    prefetch_ph(ph + 2);
    RowOp b[RB_CPL][RB_JH];
#pragma unroll
    for (int c = 0; c < RB_CPL; ++c)
#pragma unroll
      for (int jj = 0; jj < RB_JH; ++jj) b[c][jj] = Bc[ph & 1][jj < jn ? jj : 0][ql + RB_RL * c];
    bool bad_b = false;
#pragma unroll
    for (int c = 0; c < RB_CPL; ++c)
#pragma unroll
      for (int jj = 0; jj < RB_JH; ++jj) bad_b |= b[c][jj].rcp == 0.f;
    const bool any_unsafe = unsafe_a != 0 || __ballot(bad_b) != 0ull;   // wave-uniform: shared-reciprocal divide or IEEE divide
#ifdef SE3D_LM_STAMPS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    RB_STAMP(6)   // prefetch issue + det_b operands LDS -> registers
    if (nrows_left > 0) {
      if (any_unsafe) se3d_rows_compute_slow(A, Bc[ph & 1], ql, jbase, jn, row0, m2, M1, acc);
      else if (jn == RB_JH) se3d_rows_compute<RB_JH>(A, b, jbase, row0, nrows_left, acc);
      else se3d_rows_compute<RB_JLAST>(A, b, jbase, row0, nrows_left, acc);
    }
    RB_STAMP(7)   // items
    se3d_lds_barrier();   // phase ph + 1 is staged; nobody reads Bc[ph & 1] any more (its next writer is the staging of phase ph + 2)
    RB_STAMP(8)   // barrier wait
  }
  flush(((nb - 1) / RB_COLS) * RB_COLS);
  RB_STAMP(9)   // last flush
}

// =============================================================================================================
// row G (cost) — C(hyp, det) of calcCost for sweep step `b` (S3D:616-626 -> 335-390) gathered from the pair table in the
// reference's accumulation order (ascending attached view).  One workgroup per (hypothesis, frame), lanes over dets.
// =============================================================================================================
// slot_shift: column chunks (D.d.pair_chunks): pair (a, b) sits in slot se3d_pair_col_index(a, b) + slot_shift of the frame's pair slots
__global__ __launch_bounds__(128) void se3d_k_sweep_cost(Se3dDev D, int b, int slot_shift) {
  const int h = blockIdx.x, f = blockIdx.y, tid = threadIdx.x;
  const int N = D.d.N, P = D.d.P, H = D.d.H, np = D.d.pair_stride;
  if (D.nraw[f * N + b] == 0) return;
  const int nh = D.nhyp[f];
  if (nh == 0 || h >= nh) return;
  const int nd = D.nvalid[f * N + b];
  if (nd == 0) return;
  // compacted list of the views attached to this hypothesis (ascending camera = the reference's obs order)
  __shared__ unsigned long long s_off[SE3D_MAX_CAMS];  // element offset of the pair-table row (without q)
  __shared__ unsigned char s_scored[SE3D_MAX_CAMS];    // hyp.score[obs] > 0.5f
  __shared__ int s_m;
  const int32_t *arow = D.assoc + ((size_t)f * H + h) * N;
  if (tid < 64) {  // wave 0 compacts in order with ballots
    int m = 0;
    for (int a0 = 0; a0 < b; a0 += 64) {
      const int a = a0 + tid;
      const int k = (a < b) ? arow[a] : -1;
      const unsigned long long bl = __ballot(k >= 0);
      if (k >= 0) {
        const int pos = m + __popcll(bl & ((1ull << tid) - 1ull));
        const int slot = D.d.pair_chunks ? se3d_pair_col_index(a, b) + slot_shift : se3d_pair_index(a, b, N);
        s_off[pos] = ((unsigned long long)slot * P + k) * P;
        s_scored[pos] = D.pscore[((size_t)(f * N + a)) * P + k] > 0.5f ? 1 : 0;
      }
      m += __popcll(bl);
    }
    if (tid == 0) s_m = m;
  }
  __syncthreads();
  const int M = s_m;
  const int n_obs = D.nviews[f * H + h];  // == M
  const double max_epi = D.d.max_epi;
  const double tolerance = 1.0 - 1.0 / (2 * n_obs), veto_delta = 1.0 / n_obs;
  const double *tab = D.pair + (size_t)f * np * P * P;
  for (int q = tid; q < nd; q += blockDim.x) {
    double total = 0.0;
    int nused = 0, cnt = 0;
    for (int m0 = 0; m0 < M; m0 += 8) {  // 8 independent loads in flight, then the ordered fp64 accumulation
      double c[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) c[u] = (m0 + u < M) ? tab[s_off[m0 + u] + q] : -1.0;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (c[u] >= 0.0) {  // n_joints_used > 0 (padding entries are -1 as well)
          total += c[u];
          ++nused;
          if (c[u] > max_epi && (s_scored[m0 + u] || n_obs == 1)) ++cnt;
        }
      }
    }
    bool veto = false;
    if (cnt > 0) {  // tmp_veto accumulates veto_delta once per offending view (S3D:371-381)
      double tmp = 0.0;
      for (int i = 0; i < cnt; ++i) tmp += veto_delta;
      veto = tmp > tolerance;
    }
    double C;
    if (nused > 0) C = total / nused;
    else { C = MAX_COSTS; veto = true; }
    D.C[(size_t)f * H * P + h + (size_t)H * q] = C;
    D.mask[(size_t)f * H * P + h + (size_t)H * q] = (!veto && C < max_epi) ? 1 : 0;
  }
}

// =============================================================================================================
// row G (assignment) + row M — one workgroup per frame: seeding (S3D:565-586), feasibility mask bookkeeping,
// Hungarian only when ambiguous (S3D:628-634), attach / spawn (S3D:636-673).
// =============================================================================================================
__device__ __forceinline__ int se3d_block_excl_scan(bool flag, int *wsum /*LDS[5]*/, int &total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long b = __ballot(flag);
  const int within = __popcll(b & ((1ull << lane) - 1ull));
  if (lane == 0) wsum[wave] = __popcll(b);
  __syncthreads();
  int off = 0;
  for (int w = 0; w < wave; ++w) off += wsum[w];
  total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  __syncthreads();
  return off + within;
}

__device__ __forceinline__ void se3d_sweep_assign_body(const Se3dDev &D, int b, int f, double *s_dynW, int dyn_lds_bytes) {  // blockDim.x == 256
  const int tid = threadIdx.x;
  const int N = D.d.N, P = D.d.P, H = D.d.H;
  if (D.nraw[f * N + b] == 0) return;  // camera without any detection is not part of the sweep (S3D:538-555)
  const int nh = D.nhyp[f];
  const int nd = D.nvalid[f * N + b];
  int32_t *assoc = D.assoc + (size_t)f * H * N;
  __syncthreads();  // every wave has read nhyp before thread 0 updates it below
  if (nh == 0) {  // seeding from the first camera with a valid person (S3D:565-586)
    for (int k = tid; k < nd && k < H; k += 256) { assoc[(size_t)k * N + b] = k; D.nviews[f * H + k] = 1; }
    if (tid == 0) D.nhyp[f] = min(nd, H);
    return;
  }
  if (nd == 0) return;
  __shared__ MunkresShared S;
  __shared__ int assign[SE3D_MUNKRES_MAX];
  __shared__ int spawn[SE3D_MUNKRES_MAX];
  __shared__ int handled[SE3D_MUNKRES_MAX];
  __shared__ int s_amb;
  __shared__ int wsum[4];
  const uint8_t *mask = D.mask + (size_t)f * H * P;
  S1_STAMP_INIT
  S1_COUNT(8)
  if (tid == 0) s_amb = 0;
  for (int i = tid; i < SE3D_MUNKRES_MAX; i += 256) { handled[i] = 0; spawn[i] = -1; assign[i] = -1; }
  __syncthreads();
  for (int h = tid; h < nh; h += 256) {  // row counts + provisional assignment "last feasible det wins" (S3D:620-623)
    int cnt = 0, last = -1;
    for (int q = 0; q < nd; ++q)
      if (mask[h + (size_t)H * q]) { ++cnt; last = q; }
    assign[h] = last;
    if (cnt > 1) s_amb = 1;
  }
  for (int q = tid; q < nd; q += 256) {
    int cnt = 0;
    for (int h = 0; h < nh; ++h) cnt += mask[h + (size_t)H * q];
    if (cnt > 1) s_amb = 1;
  }
  __syncthreads();
  S1_STAMP(0)  // mask row/column scans
  if (tid == 0) { atomicAdd(&D.tri_cnt[4], 1u); if (s_amb) atomicAdd(&D.tri_cnt[5], 1u); }   // statistics (se3d_get_sweep_stats): steps, steps that run the Hungarian
  if (s_amb) {  // more than one feasible association: Hungarian on the full (unmasked) C (S3D:628-634)
    S1_COUNT(9)
    const double *C = D.C + (size_t)f * H * P;
    const bool in_lds = (size_t)nh * nd * sizeof(double) <= (size_t)dyn_lds_bytes;
    int rc;
    if (in_lds) {  // working matrix in LDS (ds_* addressing)
      for (int q = 0; q < nd; ++q)
        for (int h = tid; h < nh; h += 256) s_dynW[h + nh * q] = C[h + (size_t)H * q];
      __syncthreads();
      rc = se3d_munkres_block(s_dynW, nh, nh, nd, S, assign);
    } else {       // too large: HBM workspace
      double *W = D.mwork + (size_t)f * H * P;
      for (int q = 0; q < nd; ++q)
        for (int h = tid; h < nh; h += 256) W[h + (size_t)H * q] = C[h + (size_t)H * q];
      __syncthreads();
      rc = se3d_munkres_block(W, H, nh, nd, S, assign);
    }
    if (rc != 0 && tid == 0) atomicOr(&D.flags[f], SE3D_FLAG_ASSIGNMENT_FAILED);
  }
  S1_STAMP(1)  // Hungarian (copy + Munkres)
  // attach or spawn (S3D:636-660)
  bool sp = false;
  int det = -1;
  if (tid < nh) {
    det = assign[tid];
    if (det >= 0) {
      handled[det] = 1;
      if (mask[tid + (size_t)H * det]) {
        assoc[(size_t)tid * N + b] = det;
        D.nviews[f * H + tid] += 1;
      } else {
        sp = true;
      }
    }
  }
  __syncthreads();
  int tot1, tot2;
  const int i1 = se3d_block_excl_scan(sp, wsum, tot1);
  if (sp) {
    const int nhh = nh + i1;
    if (nhh < H) { assoc[(size_t)nhh * N + b] = det; D.nviews[f * H + nhh] = 1; }
  }
  const bool un = (tid < nd) && !handled[tid];  // unassigned detections become new hypotheses (S3D:662-673)
  const int i2 = se3d_block_excl_scan(un, wsum, tot2);
  if (un) {
    const int nhh = nh + tot1 + i2;
    if (nhh < H) { assoc[(size_t)nhh * N + b] = tid; D.nviews[f * H + nhh] = 1; }
  }
  if (tid == 0) {
    const int nn = nh + tot1 + tot2;
    if (nn > H) atomicOr(&D.flags[f], SE3D_FLAG_HYP_OVERFLOW);
    D.nhyp[f] = min(nn, H);
  }
  S1_STAMP(2)  // attach / spawn
}
__global__ __launch_bounds__(256) void se3d_k_sweep_assign(Se3dDev D, int b, int dyn_lds_bytes) {
  extern __shared__ __align__(16) double s_dynW_k[];  // Munkres working matrix when it fits (dyn_lds_bytes), else HBM workspace
  se3d_sweep_assign_body(D, b, blockIdx.x, s_dynW_k, dyn_lds_bytes);
}

// Cost and veto mask of camera step b for a problem small enough that one wavefront walks the whole sweep (se3d_k_sweep_fused_wave): lane <->
// (hypothesis, detection) gathers its cost from the pair table in the reference's accumulation order (ascending attached view).
__device__ __forceinline__ void se3d_sweep_cost_small(const Se3dDev &D, int b, int f, int tid, int nthr) {
  const int N = D.d.N, P = D.d.P, H = D.d.H, np = D.d.npairs;
  if (D.nraw[f * N + b] == 0) return;
  const int nh = D.nhyp[f];
  const int nd = D.nvalid[f * N + b];
  if (nh == 0 || nd == 0) return;
  const double max_epi = D.d.max_epi;
  const double *tab = D.pair + (size_t)f * np * P * P;
  for (int idx = tid; idx < nh * nd; idx += nthr) {
    const int h = idx % nh, q = idx / nh;
    const int32_t *arow = D.assoc + ((size_t)f * H + h) * N;
    const int n_obs = D.nviews[f * H + h];
    const double tolerance = 1.0 - 1.0 / (2 * n_obs), veto_delta = 1.0 / n_obs;
    double total = 0.0;
    int nused = 0, cnt = 0;
    for (int a0 = 0; a0 < b; a0 += 8) {  // 8 independent (assoc -> pair table, score) load chains in flight, ordered accumulation
      int k[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) k[u] = (a0 + u < b) ? arow[a0 + u] : -1;
      double c[8];
      float sc[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        c[u] = -1.0; sc[u] = 0.f;
        if (k[u] >= 0) {
          c[u] = tab[((size_t)se3d_pair_index(a0 + u, b, N) * P + k[u]) * P + q];
          sc[u] = D.pscore[((size_t)(f * N + a0 + u)) * P + k[u]];
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (c[u] >= 0.0) {
          total += c[u];
          ++nused;
          if (c[u] > max_epi && (sc[u] > 0.5f || n_obs == 1)) ++cnt;
        }
      }
    }
    bool veto = false;
    if (cnt > 0) {
      double tmp = 0.0;
      for (int i = 0; i < cnt; ++i) tmp += veto_delta;
      veto = tmp > tolerance;
    }
    double C;
    if (nused > 0) C = total / nused;
    else { C = MAX_COSTS; veto = true; }
    D.C[(size_t)f * H * P + h + (size_t)H * q] = C;
    D.mask[(size_t)f * H * P + h + (size_t)H * q] = (!veto && C < max_epi) ? 1 : 0;
  }
}

// =============================================================================================================
// rows T, P, O, U — one wavefront per (hypothesis, joint).
// Null vector of the 2n x 4 DLT system = eigenvector of the smallest eigenvalue of the 4x4 Gram matrix A^T A.
// Rows are built and L2-normalised in float exactly like the reference (S3D:446-453); the Gram matrix is accumulated and
// decomposed in fp64 (cyclic Jacobi), which is far more accurate than the reference's float JacobiSVD (SURVEY §8a row T:
// any sound null-vector method is within 0.04 mm).  Leave-one-out candidates and the 4n+1 sigma points of the unscented
// transform are rank-<=4 updates of the base Gram matrix, spread over the 64 lanes.
// =============================================================================================================
__device__ __forceinline__ void se3d_eig4_smallest(const double g[10], double v[4]) {
  // g packed: 00 01 02 03 11 12 13 22 23 33
  double a[4][4] = {{g[0], g[1], g[2], g[3]}, {g[1], g[4], g[5], g[6]}, {g[2], g[5], g[7], g[8]}, {g[3], g[6], g[8], g[9]}};
  double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
#define SE3D_ROT(p, q)                                                                        \
  {                                                                                           \
    const double apq = a[p][q];                                                               \
    const double scale_ = fabs(a[p][p]) + fabs(a[q][q]);                                      \
    if (fabs(apq) > 1e-40 * scale_ && apq != 0.0) {                                           \
      const double theta = (a[q][q] - a[p][p]) / (2.0 * apq);                                 \
      const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0)); \
      const double c = 1.0 / sqrt(t * t + 1.0);                                               \
      const double s = t * c;                                                                 \
      a[p][p] -= t * apq;                                                                     \
      a[q][q] += t * apq;                                                                     \
      a[p][q] = 0.0;                                                                          \
      a[q][p] = 0.0;                                                                          \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                         \
        if (r != p && r != q) {                                                               \
          const double arp = a[r][p], arq = a[r][q];                                          \
          a[r][p] = c * arp - s * arq;                                                        \
          a[p][r] = a[r][p];                                                                  \
          a[r][q] = s * arp + c * arq;                                                        \
          a[q][r] = a[r][q];                                                                  \
        }                                                                                     \
      }                                                                                       \
      _Pragma("unroll") for (int r = 0; r < 4; ++r) {                                         \
        const double vrp = V[r][p], vrq = V[r][q];                                            \
        V[r][p] = c * vrp - s * vrq;                                                          \
        V[r][q] = s * vrp + c * vrq;                                                          \
      }                                                                                       \
    }                                                                                         \
  }
  for (int sweep = 0; sweep < 12; ++sweep) {
    const double off = a[0][1] * a[0][1] + a[0][2] * a[0][2] + a[0][3] * a[0][3] + a[1][2] * a[1][2] + a[1][3] * a[1][3] + a[2][3] * a[2][3];
    const double tr = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]) + fabs(a[3][3]);
    if (off <= 1e-34 * tr * tr) break;
    SE3D_ROT(0, 1) SE3D_ROT(0, 2) SE3D_ROT(0, 3) SE3D_ROT(1, 2) SE3D_ROT(1, 3) SE3D_ROT(2, 3)
  }
#undef SE3D_ROT
  int m = 0;
  double dm = a[0][0];
  if (a[1][1] < dm) { dm = a[1][1]; m = 1; }
  if (a[2][2] < dm) { dm = a[2][2]; m = 2; }
  if (a[3][3] < dm) { dm = a[3][3]; m = 3; }
  v[0] = (m == 0) ? V[0][0] : (m == 1) ? V[0][1] : (m == 2) ? V[0][2] : V[0][3];
  v[1] = (m == 0) ? V[1][0] : (m == 1) ? V[1][1] : (m == 2) ? V[1][2] : V[1][3];
  v[2] = (m == 0) ? V[2][0] : (m == 1) ? V[2][1] : (m == 2) ? V[2][2] : V[2][3];
  v[3] = (m == 0) ? V[3][0] : (m == 1) ? V[3][1] : (m == 2) ? V[3][2] : V[3][3];
}

// 1/x to ~1 ulp (two Newton steps) or ~2^-46 (one) from the hardware seed v_rcp_f64; the inverse iteration below only needs scale factors,
// not correctly rounded quotients.
template <int NEWTON = 2>
__device__ __forceinline__ double se3d_rcp64(double x) {
  double y = __builtin_amdgcn_rcp(x);
#pragma unroll
  for (int k = 0; k < NEWTON; ++k) y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
  return y;
}

// Dehomogenised null vector of a 4x4 SPD Gram matrix by inverse iteration on its LDL^T factorisation (same eigenvector as the Jacobi solver
// above, ~15x fewer fp64 operations; the main solves start from (0, 0, 0, 1), the sigma points of the unscented transform — which perturb
// one or two rows of the base system — from the centre's solution, ~1e-3 away; the contraction per step is lambda_4 / lambda_3 << 1).
// The iterate is carried WITHOUT normalising it.  What every caller uses is the dehomogenised point w[0..2] / w[3], which is scale-free, so
// the unit-length iterate, its sign fix and an L1 test on unit vectors (18 of 42 instructions per step in the first half of round 2) are
// replaced by the point itself: p_k = w_k / w_k[3] after every step, exit when two successive points agree to tol x (1 + |p|_1).  The first
// step can pass no useful test and is taken untested.  |w| grows by ~1 / lambda_4 per step (<= 1e30 / trace with the pivot floor below);
// from the third step on the iterate is rescaled by the last pivot, so eight steps stay far inside the fp64 range.
// Returns false (caller: Jacobi fallback) when the steps do not settle or the point is not finite (w[3] = 0: point at infinity).
// NW: Newton steps on the hardware reciprocal seeds (1: ~2^-46, ample for a float32 sample; 2: ~1 ulp, the main solves).
// (tests/test_gpu_dlt.py::test_ut_covariance_error_budget prints the samples' resulting error against the exact evaluation.)
#define SE3D_SIGMA_PTOL 5e-6   // samples: |dp|_1 < 5e-6 (1 + |p|_1), i.e. an update below ~2.5e-5 m; the accepted iterate is one contraction (<= 1.5e-2) better: < 4e-7 m.
                               // (5e-7: same covariances to all printed digits, 2 % slower; 3e-5: the outlier scenes' covariance error grows from 7e-5 to 5e-4)
#define SE3D_MAIN_PTOL 1e-10   // main solves: update below 1e-10 (1 + |p|_1); the accepted iterate is one contraction (lambda_4 / lambda_3 <= ~1e-2) better, i.e.
                               // <= 1e-12 and typically 1e-14 .. 1e-16: converged for a point that is published as float32 (1e-13 cost one more step for nothing)
// MAXIT: steps allowed.  The main solves get 12 (round 5): a joint seen by ~120 cameras that carries one or two mismatched views has
// lambda_4 / lambda_3 ~ 0.05 and needs 9 - 10 steps from (0, 0, 0, 1) to 1e-10; with 8 its solve counted as not settled and the joint — well
// conditioned, kappa ~ 50 — took the float32 second pass: 59 of the 276 listed joints of four 256cam128p frames (profiles/r5/README.md section 7).
template <int NW = 1, int MAXIT = 8>
__device__ __forceinline__ bool se3d_null_invit_point(const double g[10], const double v0[4], double p[3], const double tol = SE3D_SIGMA_PTOL, double *piv = nullptr /* [4]: the LDL^T pivots */) {
  const double tr = g[0] + g[4] + g[7] + g[9];
  const double tiny = 1e-30 * tr;
  double d0 = g[0];
  if (!(d0 > tiny)) return false;
  const double i0 = se3d_rcp64<NW>(d0);
  const double l10 = g[1] * i0, l20 = g[2] * i0, l30 = g[3] * i0;
  double d1 = __builtin_fma(-l10, g[1], g[4]);
  if (!(d1 > tiny)) return false;
  const double i1 = se3d_rcp64<NW>(d1);
  const double t21 = __builtin_fma(-l20, g[1], g[5]), t31 = __builtin_fma(-l30, g[1], g[6]);
  const double l21 = t21 * i1, l31 = t31 * i1;
  double d2 = __builtin_fma(-l21, t21, __builtin_fma(-l20, g[2], g[7]));
  if (!(d2 > tiny)) return false;
  const double i2 = se3d_rcp64<NW>(d2);
  const double t32 = __builtin_fma(-l31, t21, __builtin_fma(-l30, g[2], g[8]));
  const double l32 = t32 * i2;
  double d3 = __builtin_fma(-l32, t32, __builtin_fma(-l31, t31, __builtin_fma(-l30, g[3], g[9])));
  if (!(d3 > tiny)) d3 = tiny;  // numerically singular: the solve still returns the null direction
  if (piv) { piv[0] = d0; piv[1] = d1; piv[2] = d2; piv[3] = d3; }
  const double i3 = se3d_rcp64<NW>(d3);
  double x0 = v0[0], x1 = v0[1], x2 = v0[2], x3 = v0[3];
  double q0 = 0.0, q1 = 0.0, q2 = 0.0;
  bool conv = false;
  for (int it = 0; it < MAXIT; ++it) {
    const double y0 = x0, y1 = __builtin_fma(-l10, y0, x1), y2 = __builtin_fma(-l21, y1, __builtin_fma(-l20, y0, x2)),
                 y3 = __builtin_fma(-l32, y2, __builtin_fma(-l31, y1, __builtin_fma(-l30, y0, x3)));
    const double w3 = y3 * i3, w2 = __builtin_fma(-l32, w3, y2 * i2), w1 = __builtin_fma(-l31, w3, __builtin_fma(-l21, w2, y1 * i1)),
                 w0 = __builtin_fma(-l30, w3, __builtin_fma(-l20, w2, __builtin_fma(-l10, w1, y0 * i0)));
    const double iw = se3d_rcp64<NW>(w3);
    const double p0 = w0 * iw, p1 = w1 * iw, p2 = w2 * iw;
    if (it >= 1) {
      const double e = fabs(p0 - q0) + fabs(p1 - q1) + fabs(p2 - q2);
      if (e < tol * (1.0 + fabs(p0) + fabs(p1) + fabs(p2))) { q0 = p0; q1 = p1; q2 = p2; conv = true; break; }
    }
    q0 = p0; q1 = p1; q2 = p2;
    const double sc = it >= 1 ? d3 : 1.0;
    x0 = w0 * sc; x1 = w1 * sc; x2 = w2 * sc; x3 = w3 * sc;
  }
  p[0] = q0; p[1] = q1; p[2] = q2;
  return conv;
}

// Dehomogenised null vector of g: the inverse iteration above, the Jacobi solver when it does not settle.
template <int NW = 1>
__device__ __forceinline__ void se3d_solve_point(const double g[10], const double v0[4], double p[3], const double tol = SE3D_SIGMA_PTOL) {
  if (se3d_null_invit_point<NW>(g, v0, p, tol)) return;
  double e4[4];
  se3d_eig4_smallest(g, e4);
  const double iw = se3d_rcp64<NW>(e4[3]);
  p[0] = e4[0] * iw; p[1] = e4[1] * iw; p[2] = e4[2] * iw;
}

// ---- which joints leave the fp64 fast path (round 4) -------------------------------------------------------------------------------------
// The reference solves every DLT system in float32 (S3D:456); the fp64 solves here agree with it to c 2^-24 kappa metres,
// kappa = (1 + |X|^2) s1 / (s3 - s4) over the singular values of the 2n x 4 system (c <= 4.9 over 3e5 joints, profiles/r3/cond_study.txt).
// se3d_kappa_flag tests an UPPER bound of kappa from what a lane has in registers — the Gram matrix g and the solution p:
//   lambda4 <= Rayleigh quotient of (p, 1);  s1 <= sqrt(trace);  lambda3 >= the smallest root of the characteristic polynomial deflated by
//   lambda4, a cubic with three positive roots, approached from below (b3 / b2 <= lambda3 because 1/l1 + 1/l2 + 1/l3 = b2 / b3; left of its
//   smallest root the cubic is increasing and concave, so Newton steps from there never pass the root).
// Measured on 3.8e5 joints (tools/study/kappa_bound_study.py, profiles/r4/kappa_bound_study.txt): bound / kappa between 1.01 and 1.7 (median
// 1.25) with two Newton steps.  A joint whose bound reaches SE3D_EXACT_KAPPA is redone by se3d_k_triangulate_exact with the oracle's restatement of the reference's float32 arithmetic; the
// worst float32-vs-fp64 deviation among the joints that stay here was 0.055 mm in that study (the contract is 1 mm).
#define SE3D_EXACT_KAPPA 1.0e3
// Decisions of the outlier rejection (S3D:748-844) compare the reprojection error with g_reproj_error_max_acceptable; an error within this
// relative band below the threshold (or above it) is re-evaluated in float32 too: the two solvers' points differ by < 0.1 mm on a joint that
// passed the kappa test, i.e. by < 1e-3 of 0.05 in the error for a camera 2 m away.
#define SE3D_EXACT_ERR_GUARD 0.01
// A joint whose unscented-transform covariance comes out with a variance above (1 m)^2 is redone in float32 as well: its sigma samples land
// metres to tens of metres from where the rays converge, THOSE systems are ill conditioned whatever the base system's kappa is (seen in round 3
// on spurious persons of crowded scenes: base kappa 59 - 1150, sigma 30 - 50 m, float32 and fp64 covariances 1.3e-3 - 7.9e-3 apart).
#define SE3D_EXACT_VAR_MAX 1.0
// Evaluated in float32 (half the issue cost of fp64 on this chip): only lambda3 ~ 1e-2 .. 1e-4 of the trace and the gap to lambda4 matter, the
// threshold is met with a margin of 18 x, and the study above gives the same flags for the float32 and the fp64 evaluation (lambda3 starts
// from the second-order bound b3 / sqrt(b2^2 - 2 b1 b3) <= lambda3 — 1/l1^2 + 1/l2^2 + 1/l3^2 = (b2^2 - 2 b1 b3) / b3^2; bound / kappa 1.04 .. 2.1,
// median 1.4 — and takes two Newton steps from there: 1.01 .. 1.7, median 1.25).  ~110 float operations, no division by the gap:
// kappa < K  <=>  (1 + |p|^2)^2 trace < K^2 (s3 - s4)^2.
// Returns true when the joint must go to the second pass.
__device__ __forceinline__ bool se3d_kappa_flag(const float g[10], float p0, float p1, float p2) {
  // g packed: 00 01 02 03 11 12 13 22 23 33
  const float a = g[0], b = g[1], c = g[2], d = g[3], e = g[4], f = g[5], h = g[6], i = g[7], k = g[8], l = g[9];
  const float c1 = (a + e) + (i + l);
  const float m12 = fmaf(e, i, -f * f), m13 = fmaf(e, l, -h * h), m23 = fmaf(i, l, -k * k);
  const float c2 = (fmaf(a, e, -b * b) + fmaf(a, i, -c * c)) + (fmaf(a, l, -d * d) + m12) + (m13 + m23);
  const float M012 = fmaf(c, fmaf(b, f, -c * e), fmaf(-b, fmaf(b, i, -c * f), a * m12));
  const float M013 = fmaf(d, fmaf(b, h, -d * e), fmaf(-b, fmaf(b, l, -d * h), a * m13));
  const float M023 = fmaf(d, fmaf(c, k, -d * i), fmaf(-c, fmaf(c, l, -d * k), a * m23));
  const float M123 = fmaf(h, fmaf(f, k, -h * i), fmaf(-f, fmaf(f, l, -h * k), e * m23));
  const float c3 = (M012 + M013) + (M023 + M123);
  // Rayleigh quotient of v = (p, 1): an upper bound of lambda4
  const float w0 = fmaf(a, p0, fmaf(b, p1, fmaf(c, p2, d)));
  const float w1 = fmaf(b, p0, fmaf(e, p1, fmaf(f, p2, h)));
  const float w2 = fmaf(c, p0, fmaf(f, p1, fmaf(i, p2, k)));
  const float w3 = fmaf(d, p0, fmaf(h, p1, fmaf(k, p2, l)));
  const float vv = fmaf(p0, p0, fmaf(p1, p1, fmaf(p2, p2, 1.0f)));
  const float lam4 = fmaxf(fmaf(w0, p0, fmaf(w1, p1, fmaf(w2, p2, w3))) * __builtin_amdgcn_rcpf(vv), 0.f);
  const float b1 = c1 - lam4, b2 = fmaf(-lam4, b1, c2), b3 = fmaf(-lam4, b2, c3);
  const float t = fmaf(b2, b2, -2.0f * (b1 * b3));
  float x = b3 * __builtin_amdgcn_rsqf(t);                             // <= lambda3 (second-order bound)
#pragma unroll
  for (int it = 0; it < 2; ++it) {   // two Newton steps on the deflated cubic from below (left of its smallest root it is increasing and concave:
    const float q = fmaf(fmaf(x - b1, x, b2), x, -b3);                // the steps approach lambda3 without passing it): bound / kappa 1.01 .. 1.7
    const float dq = fmaf(fmaf(3.0f, x, -2.0f * b1), x, b2);
    x = fmaf(-q, __builtin_amdgcn_rcpf(dq), x);
  }
  const float gap2 = (x + lam4) - 2.0f * __builtin_amdgcn_sqrtf(x * lam4);   // (s3 - s4)^2
  const float lhs = (vv * vv) * c1;
  const bool fine = t > 0.f && b3 > 0.f && x > lam4 && lhs < (float)(SE3D_EXACT_KAPPA * SE3D_EXACT_KAPPA) * gap2;
  return !fine;
}
// Level 1 of the conditioning test: a SUFFICIENT condition for "both systems of the joint are well conditioned" from what the weighted solve
// leaves in registers — twelve fp64 multiplications, no division, no square root — so that se3d_kappa_flag (~90 operations: 4.8 % of the
// hypothesis kernel when every joint took it, profiles/r4/README.md) runs only for the joints that fail it (1.4 % on the benchmark's scenes,
// 13 % on the randomised sweep's).  g: weighted Gram matrix, piv: its LDL^T pivots, p: the solution, n: usable views (confidences in
// [0.30, 1]: S3D:725 and the caller's check).  With B the leading 3 x 3 block (eigenvalues mu1 >= mu2 >= mu3, interlacing lambda3 >= mu3):
//   lambda3(Gw) >= mu3 >= det B / (mu1 mu2) >= 4 d0 d1 d2 / tr(B)^2;     lambda4(Gw) <= x^T Gw x / |x|^2 at x = (p*, 1) = d3 / (1 + |p*|^2);
//   Gu - Gw = sum (1 - c_i^2) row row^T >= 0 and Gu <= Gw / 0.09:  lambda3(Gu) >= lambda3(Gw), lambda4(Gu) <= 11.2 lambda4(Gw), tr(Gu) = 2 n.
// If 200 d3 / vv <= lambda3_lb (so that s4 <= s3 / 4 for both systems, 10 % slack for p* vs p) then (s3 - s4)^2 >= 9/16 lambda3_lb and
// kappa^2 <= vv^2 2n (16/9) / lambda3_lb for both.
__device__ __forceinline__ bool se3d_cond_level1_ok(const double g[10], const double piv[4], const double p[3], int n) {
  const double tr3 = (g[0] + g[4]) + g[7];
  const double L4 = 4.0 * (piv[0] * piv[1]) * piv[2], T2 = tr3 * tr3;          // lambda3_lb = L4 / T2
  const double vv = __builtin_fma(p[0], p[0], __builtin_fma(p[1], p[1], __builtin_fma(p[2], p[2], 1.0)));
  const bool gap_ok = 200.0 * piv[3] * T2 <= L4 * vv;
  const bool kap_ok = (vv * vv) * ((32.0 / 9.0) * (double)n) * T2 < (SE3D_EXACT_KAPPA * SE3D_EXACT_KAPPA) * L4;
  return gap_ok && kap_ok;
}
// hands (frame, hypothesis, joint) to the second pass (se3d_k_triangulate_exact / _slow)
__device__ __forceinline__ void se3d_tri_list_push(const Se3dDev &D, int f, int h, int j) {
  const uint32_t at = atomicAdd(&D.tri_cnt[0], 1u);
  D.tri_list[at] = ((uint32_t)(f * D.d.H + h) << 5) | (uint32_t)j;
}

// 1.0f / sqrtf(z), both IEEE: from se3d_sqrt_rcp_core inside its range, else the expressions themselves
__device__ __forceinline__ float se3d_rsqrt_exact(float z) {
  const unsigned u = __float_as_uint(z);
  if (u >= SE3D_EXACT_LO && u <= SE3D_EXACT_HI) {
    float inv;
    const float root = se3d_sqrt_rcp_core(z, inv);
    if (!se3d_all_ones(root)) return inv;
  }
  return 1.0f / sqrtf(z);
}
// rows of the DLT system for one view: r0 = normalize(x*P2 - P0), r1 = normalize(y*P2 - P1)  (float, S3D:446-449)
__device__ __forceinline__ void se3d_dlt_rows(const float *__restrict__ M, float x, float y, float r0[4], float r1[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) { r0[k] = x * M[8 + k] - M[k]; r1[k] = y * M[8 + k] - M[4 + k]; }
  // L2 normalisation (S3D:448-449): one correctly rounded reciprocal per row and four multiplies instead of four IEEE divides
  // (<= 1 ulp of float per component, 5 orders of magnitude below the 1 mm parity tolerance on the triangulated point)
  // (round 5: 1.0f / sqrtf(z) from one v_rsq_f32 where se3d_sqrt_rcp_core is the IEEE result — z in [2^-60, 2^60], root not all ones)
  const float z0 = (r0[0] * r0[0] + r0[1] * r0[1]) + (r0[2] * r0[2] + r0[3] * r0[3]);
  if (z0 > 0.f) { const float inv = se3d_rsqrt_exact(z0);
#pragma unroll
    for (int k = 0; k < 4; ++k) r0[k] = r0[k] * inv; }
  const float z1 = (r1[0] * r1[0] + r1[1] * r1[1]) + (r1[2] * r1[2] + r1[3] * r1[3]);
  if (z1 > 0.f) { const float inv = se3d_rsqrt_exact(z1);
#pragma unroll
    for (int k = 0; k < 4; ++k) r1[k] = r1[k] * inv; }
}

template <bool SUB = false>
__device__ __forceinline__ void se3d_gram_add(double g[10], const float r[4]) {
  const double a = r[0], b = r[1], c = r[2], d = r[3];
  const double sa = SUB ? -a : a, sb = SUB ? -b : b, sc = SUB ? -c : c, sd = SUB ? -d : d;
  g[0] = __builtin_fma(sa, a, g[0]); g[1] = __builtin_fma(sa, b, g[1]); g[2] = __builtin_fma(sa, c, g[2]); g[3] = __builtin_fma(sa, d, g[3]);
  g[4] = __builtin_fma(sb, b, g[4]); g[5] = __builtin_fma(sb, c, g[5]); g[6] = __builtin_fma(sb, d, g[6]);
  g[7] = __builtin_fma(sc, c, g[7]); g[8] = __builtin_fma(sc, d, g[8]);
  g[9] = __builtin_fma(sd, d, g[9]);
}

// ---- sigma samples of the unscented transform (S3D:471-523) --------------------------------------------------------------------
// A view contributes four samples: (-dx1, -dy1), (0, -dy2), (+dx1, +dy1), (0, +dy2).  Row r0 = normalize(x P2 - P0) of the DLT system
// depends on x only and r1 = normalize(y P2 - P1) on y only, so the two "y-only" samples leave r0 untouched: class A (centre + the
// 2n diagonal samples) rebuilds both rows, class B (the 2n y-only samples) rebuilds r1 alone.  Sample rows are normalised with the
// hardware reciprocal square root (1 ulp): the samples feed a covariance that is compared at 1e-3, not the association or the point (the
// weighted solve keeps the exact row build of se3d_dlt_rows).
__device__ __forceinline__ void se3d_dlt_row_fast(const float *__restrict__ M, int which, float coord, float r[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) r[k] = coord * M[8 + k] - M[4 * which + k];
  const float z = (r[0] * r[0] + r[1] * r[1]) + (r[2] * r[2] + r[3] * r[3]);
  if (z > 0.f) { const float inv = __builtin_amdgcn_rsqf(z);
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = r[k] * inv; }
}
// Gram matrix of sample (view c, kind): kind 0 centre (unchanged), 1 / 2 class A with sign -1 / +1, 3 / 4 class B with sign -1 / +1.
__device__ __forceinline__ void se3d_sigma_gram(double g[10], const float *__restrict__ M, float x, float y, float l11, float l21, float l22, float bsc,
                                                const float o0[4], const float o1[4], int kind) {
  if (kind == 0) return;
  const float sgn = (kind & 1) ? -1.f : 1.f;
  float n1[4];
  if (kind <= 2) {
    float n0[4];
    se3d_dlt_row_fast(M, 0, x + sgn * (l11 * bsc), n0);
    se3d_dlt_row_fast(M, 1, y + sgn * (l21 * bsc), n1);
    se3d_gram_add<true>(g, o0); se3d_gram_add<true>(g, o1);
    se3d_gram_add(g, n0); se3d_gram_add(g, n1);
  } else {
    se3d_dlt_row_fast(M, 1, y + sgn * (l22 * bsc), n1);
    se3d_gram_add<true>(g, o1);
    se3d_gram_add(g, n1);
  }
}

__device__ __forceinline__ double se3d_readlane_f64(double v, int l) {  // l uniform
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double se3d_wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// reprojection error of float point X in one view (S3D:430-433)
__device__ __forceinline__ float se3d_reproj1(const float *__restrict__ M, float X, float Y, float Z, float px, float py) {
  const float hx = (M[0] * X + (M[1] * Y + M[2] * Z)) + M[3];
  const float hy = (M[4] * X + (M[5] * Y + M[6] * Z)) + M[7];
  const float hz = (M[8] * X + (M[9] * Y + M[10] * Z)) + M[11];
  // Hardware reciprocal and square root (1 ulp each) instead of two IEEE divisions and an IEEE root (round 5): this error only meets thresholds —
  // the outlier gate with its 1 % guard band (SE3D_EXACT_ERR_GUARD: everything within it goes to the float32 second pass, which forms its own
  // errors with the reference's expressions) — and never reaches a published number.  hz = 0: infinite / NaN error, i.e. the second pass.
  const float iz = __builtin_amdgcn_rcpf(hz);
  const float dx = hx * iz - px, dy = hy * iz - py;
  return __builtin_amdgcn_sqrtf(dx * dx + dy * dy);
}

// The 2x2 keypoint covariance in normalised image coordinates and its lower Cholesky factor (S3D:476-480).  They feed only the sigma offsets
// of the unscented transform, whose covariance is compared at 1e-3: hardware reciprocal / square root (1 ulp each) instead of four IEEE
// division and two IEEE square-root sequences per (joint, view).  A factor that is not finite (non-SPD input) stays not finite.
__device__ __forceinline__ void se3d_norm_cov2(const float cov[3], float fx, float fy, float &k00, float &k01, float &k11) {
  const float ifx = __builtin_amdgcn_rcpf(fx), ify = __builtin_amdgcn_rcpf(fy);
  k00 = cov[0] * (ifx * ifx); k01 = cov[1] * (ifx * ify); k11 = cov[2] * (ify * ify);
}
__device__ __forceinline__ void se3d_chol2(float k00, float k01, float k11, float &l11, float &l21, float &l22) {
  l11 = __builtin_amdgcn_sqrtf(k00);
  l21 = k01 * __builtin_amdgcn_rcpf(l11);
  l22 = __builtin_amdgcn_sqrtf(k11 - l21 * l21);
}
// Row N on the fly (S3D:312-333): keypoint j of the detection in FILTERED slot k of camera c, read from the caller's raw Person2D payload
// and normalised in registers with the expressions of se3d_k_normalize (bit-identical values; no normalised copy in HBM).
// Returns conf >= 0.30 (the triangulation gate, S3D:724-732); kp = (x, y, conf), kc = (a00, a01, a11) only when it does.
__device__ __forceinline__ bool se3d_load_norm(const Se3dDev &D, int f, int c, int k, int j, float4 &kp, float4 &kc) {
  const size_t base = ((size_t)(f * D.d.N + c)) * D.d.P;
  const se3d_keypoint2d r = D.raw[base + D.orig[base + k]].keypoints[j];
  kp = make_float4(0.f, 0.f, -1.f, 0.f);
  kc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (!(r.score >= TRI_THRESH)) return false;
  const float fx = D.K[c * 4 + 0], fy = D.K[c * 4 + 1], cx = D.K[c * 4 + 2], cy = D.K[c * 4 + 3];
  kp = make_float4((r.x - cx) / fx, (r.y - cy) / fy, r.score, 0.f);
  se3d_norm_cov2(r.cov, fx, fy, kc.x, kc.y, kc.z);
  return true;
}
// The triangulation kernels write EVERY field of the hypothesis slots they own (no zero-fill pass before them).
__device__ __forceinline__ void se3d_zero_kp_words(se3d_keypoint3d *kp, int lane, int nlanes) {
  uint32_t *w = reinterpret_cast<uint32_t *>(kp);
  for (int i = lane; i < (int)(sizeof(se3d_keypoint3d) / 4); i += nlanes) w[i] = 0u;
}
// header (id, score) and the fusion slots no input joint maps to (simple: Neck, MidHip, Head, Belly; h36m: eyes and ears)
__device__ __forceinline__ void se3d_zero_unused_slots(se3d_person3d &p, int pm, int lane) {
  uint32_t used = 0u;
#pragma unroll
  for (int j = 0; j < SE3D_NKP; ++j) used |= 1u << c_kp2fusion[pm][j];
  if (lane == 0) { p.id = 0u; p.score = 0.f; }
  for (uint32_t m = ~used & ((1u << SE3D_NFU) - 1u); m; m &= m - 1u) se3d_zero_kp_words(&p.keypoints[__ffs((int)m) - 1], lane, 64);
}

struct TriView {  // per-wave view list in LDS (structure of arrays, stride = N)
  float *x, *y, *cf, *ca, *cb, *cc;
  float *r0[4], *r1[4];
  int *cam;
};

// One (hypothesis, joint): rows T, P, O, U for a single joint by one wavefront.  `base`: 15 N floats of LDS owned by the wavefront,
// `sRt`: the rig's camera matrices in LDS.
__device__ __forceinline__ void se3d_tri_joint_body(const Se3dDev &D, int f, int h, int j, int lane, float *base, const float *sRt) {
  const int N = D.d.N, H = D.d.H;
  TriView T;
  T.x = base; T.y = base + N; T.cf = base + 2 * N; T.ca = base + 3 * N; T.cb = base + 4 * N; T.cc = base + 5 * N;
  for (int k = 0; k < 4; ++k) { T.r0[k] = base + (6 + k) * N; T.r1[k] = base + (10 + k) * N; }
  T.cam = reinterpret_cast<int *>(base + 14 * N);

  // ---- gather the views that see this joint with conf >= 0.30 (S3D:724-732), ascending camera order ----
  const int32_t *arow = D.assoc + ((size_t)f * H + h) * N;
  int n = 0;
  for (int c0 = 0; c0 < N; c0 += 64) {
    const int c = c0 + lane;
    bool ok = false;
    float4 kp = make_float4(0, 0, -1, 0), kc = make_float4(0, 0, 0, 0);
    if (c < N) {
      const int k = arow[c];
      if (k >= 0) ok = se3d_load_norm(D, f, c, k, j, kp, kc);
    }
    const unsigned long long bl = __ballot(ok);
    if (ok) {
      const int pos = n + __popcll(bl & ((1ull << lane) - 1ull));
      T.x[pos] = kp.x; T.y[pos] = kp.y; T.cf[pos] = kp.z;
      {  // lower Cholesky factor of the view's 2x2 keypoint covariance (S3D:476-480), shared by the view's four sigma points
        float l11, l21, l22;
        se3d_chol2(kc.x, kc.y, kc.z, l11, l21, l22);
        T.ca[pos] = l11; T.cb[pos] = l21; T.cc[pos] = l22;
      }
      T.cam[pos] = c;
    }
    n += __popcll(bl);
  }
  __builtin_amdgcn_wave_barrier();  // view list written by other lanes of this wave is read below
  if (n < 2) {  // cannot triangulate (S3D:734-736): the slot is default-constructed (score 0)
    se3d_zero_kp_words(&D.hypP[(size_t)f * H + h].keypoints[c_kp2fusion[D.d.pose_method][j]], lane, 64);
    return;
  }

  // ---- rows + Gram matrices: Gw (confidence weighted, S3D:450-453) and Gu (unweighted, used by the UT) ----
  double gw_[10], gu_[10];
  // Gram entry owned by this lane in the small-n path: lanes 0-9 unweighted (00 01 02 03 11 12 13 22 23 33), 10-19 weighted
  const int ge = lane < 10 ? lane : (lane < 20 ? lane - 10 : 0);
  const int gi = (int)((0x3221110000ull >> (4 * ge)) & 0xFull), gj = (int)((0x3323213210ull >> (4 * ge)) & 0xFull);
  auto build = [&](int nn) {
    if (nn <= 24) {
      // few views: rows by the first nn lanes, then every Gram entry is accumulated by its own lane over the views (broadcast LDS
      // reads, no cross-lane reduction) and handed to all lanes with v_readlane
      if (lane < nn) {
        float r0[4], r1[4];
        se3d_dlt_rows(sRt + 12 * T.cam[lane], T.x[lane], T.y[lane], r0, r1);
#pragma unroll
        for (int k = 0; k < 4; ++k) { T.r0[k][lane] = r0[k]; T.r1[k][lane] = r1[k]; }
      }
      __builtin_amdgcn_wave_barrier();
      double acc = 0.0;
      if (lane < 20) {
        const float *ri0 = base + (6 + gi) * N, *rj0 = base + (6 + gj) * N, *ri1 = base + (10 + gi) * N, *rj1 = base + (10 + gj) * N;
        const bool weighted = lane >= 10;
        for (int v = 0; v < nn; ++v) {
          const float cf = T.cf[v];
          float a0 = ri0[v], b0 = rj0[v], a1 = ri1[v], b1 = rj1[v];
          if (weighted) { a0 = a0 * cf; b0 = b0 * cf; a1 = a1 * cf; b1 = b1 * cf; }
          acc = __builtin_fma((double)a0, (double)b0, acc);
          acc = __builtin_fma((double)a1, (double)b1, acc);
        }
      }
#pragma unroll
      for (int k = 0; k < 10; ++k) { gu_[k] = se3d_readlane_f64(acc, k); gw_[k] = se3d_readlane_f64(acc, 10 + k); }
      return;
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) { gw_[k] = 0.0; gu_[k] = 0.0; }
    for (int v = lane; v < nn; v += 64) {
      float r0[4], r1[4], w0[4], w1[4];
      se3d_dlt_rows(sRt + 12 * T.cam[v], T.x[v], T.y[v], r0, r1);
      const float cf = T.cf[v];
#pragma unroll
      for (int k = 0; k < 4; ++k) { T.r0[k][v] = r0[k]; T.r1[k][v] = r1[k]; w0[k] = r0[k] * cf; w1[k] = r1[k] * cf; }
      se3d_gram_add(gu_, r0); se3d_gram_add(gu_, r1);
      se3d_gram_add(gw_, w0); se3d_gram_add(gw_, w1);
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) { gw_[k] = se3d_wave_sum(gw_[k]); gu_[k] = se3d_wave_sum(gu_[k]); }
    __builtin_amdgcn_wave_barrier();  // rows in LDS are consumed by other lanes (leave-one-out, sigma points)
  };
  // weighted reprojection error of float point (X,Y,Z) over views [0,nn) except `skip` (S3D:425-438)
  auto reproj = [&](int nn, float X, float Y, float Z) -> double {
    double se = 0.0, sc = 0.0;
    for (int v = lane; v < nn; v += 64) {
      const float e = se3d_reproj1(sRt + 12 * T.cam[v], X, Y, Z, T.x[v], T.y[v]);
      se += (double)(T.cf[v] * e);
      sc += (double)T.cf[v];
    }
    return se3d_wave_sum(se) / se3d_wave_sum(sc);
  };
  build(n);
  double pw[3];
  const double e4start[4] = {0.0, 0.0, 0.0, 1.0};  // the null vector (X, Y, Z, 1)/|.| of a finite point always has a w component
  double piv_[4];
  const bool conv0 = se3d_null_invit_point<2, 12>(gw_, e4start, pw, SE3D_MAIN_PTOL, piv_);
  if (!conv0) {
    double e4[4];
    se3d_eig4_smallest(gw_, e4);
    const double iw = se3d_rcp64<2>(e4[3]);
    pw[0] = e4[0] * iw; pw[1] = e4[1] * iw; pw[2] = e4[2] * iw;
  }
  float X = (float)pw[0], Y = (float)pw[1], Z = (float)pw[2];
  double err = reproj(n, X, Y, Z);
  float avg_score = 0.f;   // the reference's float sum in view order and float quotient (S3D:730, 738): bit-identical scores on this path
  for (int v = 0; v < n; ++v) avg_score += T.cf[v];
  avg_score /= (float)n;
  {
    // Round 4: a joint whose fp64 solve may differ from the float32 JacobiSVD result (as the oracle restates it) by more than the contract allows — ill-conditioned
    // weighted or unweighted system, a solve that did not settle — or whose reprojection error is above / near the outlier threshold is handed
    // to se3d_k_triangulate_exact (the oracle's restatement of the reference's arithmetic, bit for bit).  Everything decided from here on is uniform over the wavefront.
    bool flag = !conv0 || !(err <= REPROJ_MAX * (1.0 - SE3D_EXACT_ERR_GUARD));
    bool hi = false;   // a confidence above 1 (the level-1 test assumes c <= 1)
    for (int v = lane; v < n; v += 64) hi |= T.cf[v] > 1.0f;
    if (!flag && (__ballot(hi) != 0ull || !se3d_cond_level1_ok(gw_, piv_, pw, n))) {   // uniform
      float gwf[10], guf[10];
#pragma unroll
      for (int k = 0; k < 10; ++k) { gwf[k] = (float)gw_[k]; guf[k] = (float)gu_[k]; }
      flag = se3d_kappa_flag(gwf, X, Y, Z) || se3d_kappa_flag(guf, X, Y, Z);
    }
    if (flag) {
      if (lane == 0) se3d_tri_list_push(D, f, h, j);
      return;
    }
  }

  // (outlier rejection, S3D:748-838, and the score's down-weighting, S3D:840-844: reprojection errors near or above the threshold went to
  // the second pass above — here err <= 0.99 g_reproj_error_max_acceptable and the score is the mean confidence)
  // ---- unscented transform (S3D:471-523): 4n+1 sigma points, each an unweighted re-triangulation ----
  // Defined policy: a kept view whose 2x2 keypoint covariance is not SPD (NaN / inf Cholesky factor, S3D:473-475) makes
  // all entries of the joint's 3-D covariance NaN; position and score are unaffected.
  bool cov_bad = false;
  for (int v = lane; v < n; v += 64) cov_bad |= !(isfinite(T.ca[v]) && isfinite(T.cb[v]) && isfinite(T.cc[v]));
  cov_bad = __ballot(cov_bad) != 0ull;
  const int dim = 2 * n;
  const float bsc = sqrtf((float)dim + 0.5f);
  const float wgt = 1.0f / (2.0f * ((float)dim + 0.5f));  // all weights equal because 2*kappa == 1
  double cv[6] = {0, 0, 0, 0, 0, 0};
  const int nsamp = 4 * n + 1;
  bool sample_bad = false;   // a sample's inverse iteration did not settle: the joint goes to the second pass
  const double vstart[4] = {(double)X, (double)Y, (double)Z, 1.0};  // warm start of the sigma-point solves: the published point (the iteration is scale-free)
  // samples 0 .. 2n: centre + class A (both rows rebuilt); 2n + 1 .. 4n: class B (y-only samples, r1 alone) — see se3d_sigma_gram
  for (int s0 = 0; s0 < nsamp; s0 += 64) {
    const int s = s0 + lane;
    if (s < nsamp) {
      double g[10];
#pragma unroll
      for (int k = 0; k < 10; ++k) g[k] = gu_[k];
      if (s > 0) {
        const bool clsB = s > 2 * n;
        const int q = clsB ? s - 2 * n - 1 : s - 1, c = q >> 1;
        float o0[4], o1[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { o0[k] = T.r0[k][c]; o1[k] = T.r1[k][c]; }
        se3d_sigma_gram(g, sRt + 12 * T.cam[c], T.x[c], T.y[c], T.ca[c], T.cb[c], T.cc[c], bsc, o0, o1, (clsB ? 3 : 1) + (q & 1));
      }
      double pt[3];
      sample_bad |= !se3d_null_invit_point(g, vstart, pt);   // (no Jacobi fallback here: a sample that does not settle sends the joint to the second pass)
      const float sx = (float)pt[0] - X, sy = (float)pt[1] - Y, sz = (float)pt[2] - Z;
      const double wx = (double)(sx * wgt), wy = (double)(sy * wgt), wz = (double)(sz * wgt);
      cv[0] += wx * sx; cv[1] += wx * sy; cv[2] += wx * sz; cv[3] += wy * sy; cv[4] += wy * sz; cv[5] += wz * sz;
    }
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) cv[k] = cov_bad ? (double)__builtin_nanf("") : se3d_wave_sum(cv[k]);
  if (!cov_bad && (__ballot(sample_bad) != 0ull || !(cv[0] < SE3D_EXACT_VAR_MAX && cv[3] < SE3D_EXACT_VAR_MAX && cv[5] < SE3D_EXACT_VAR_MAX))) {
    if (lane == 0) se3d_tri_list_push(D, f, h, j);   // wide covariance / unsettled sample: the samples' systems are ill conditioned — second pass (uniform)
    return;
  }
  if (lane == 0) {
    se3d_keypoint3d *kp = &D.hypP[(size_t)f * H + h].keypoints[c_kp2fusion[D.d.pose_method][j]];
    kp->joint[0] = (double)X; kp->joint[1] = (double)Y; kp->joint[2] = (double)Z;
    kp->score = avg_score; kp->_pad = 0.f;
    kp->cov[0] = (double)(float)cv[0]; kp->cov[1] = (double)(float)cv[1]; kp->cov[2] = (double)(float)cv[2];
    kp->cov[3] = (double)(float)cv[3]; kp->cov[4] = (double)(float)cv[4]; kp->cov[5] = (double)(float)cv[5];
  }
}

__global__ __launch_bounds__(256, 3) void se3d_k_triangulate(Se3dDev D) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int f = blockIdx.y;
  const int N = D.d.N, H = D.d.H;
  const int gw = blockIdx.x * 4 + wave;
  const int h = gw / SE3D_NKP, j = gw - h * SE3D_NKP;
  // camera matrices of the rig staged once per workgroup (every view of every wavefront reads its 3x4 row block several times)
  float *sRt = reinterpret_cast<float *>(smem) + (size_t)4 * 15 * N;
  for (int i = threadIdx.x; i < 12 * N; i += 256) sRt[i] = D.Rt[i];
  __syncthreads();
  if (h >= D.nhyp[f]) return;
  if (D.nviews[f * H + h] < 2) return;
  if (D.d.shard_world > 1 && h % D.d.shard_world != D.d.shard_rank) return;  // hypothesis shard: another rank's hypothesis
  if (j == 0) se3d_zero_unused_slots(D.hypP[(size_t)f * H + h], D.d.pose_method, lane);
  se3d_tri_joint_body(D, f, h, j, lane, reinterpret_cast<float *>(smem) + (size_t)wave * 15 * N, sRt);
}

// =============================================================================================================
// Small rigs (N <= 16): one wavefront per HYPOTHESIS instead of one per (hypothesis, joint).  The per-joint kernel above spends most
// of its 1.9 k instructions per joint on work that uses a fraction of the wavefront (row build: n <= 16 lanes; the weighted solve:
// the same 4x4 problem on all 64 lanes; 4n+1 ~ 37 sigma points on 64 lanes).  Here the 17 joints of a hypothesis share the
// wavefront: (joint, view) pairs fill the lanes for the point / row phases, the 17 weighted solves run on 17 lanes at once, and the
// sigma points of all joints are enumerated densely (17 x 37 samples = 10 rounds of 64 instead of 17 rounds).  Same device functions
// and the same per-sample arithmetic as the per-joint kernel (identical triangulated points; covariance and error sums are taken
// sequentially instead of by a butterfly).  Joints that need outlier rejection (reprojection error above the threshold with >= 3
// views, S3D:748-838) are rare and go through the per-joint routine afterwards.
// =============================================================================================================
#define TH_MAXV 16
// A hypothesis is split into TH_PARTS one-wavefront workgroups of 6 + 6 + 5 joints (round 1 / first half of round 2: two of 9 + 8).  Three
// parts: 8 KB of LDS per wavefront instead of 11.6 KB, so that LDS no longer caps a CU at 13 wavefronts; with the kernel compiled for four
// wavefronts per SIMD (128 VGPRs, 56 B of scratch outside the sample loop) 16 are resident instead of 12; and 10 / 12 lanes per joint
// walk the 4n + 1 sigma samples of a 12-view joint in 3 + 3 trips instead of 4 + 4.  13.09 -> 12.38 (three parts) -> 11.50 (128 VGPRs, 56 B of
// scratch per lane: 5.5 GB of HBM traffic per launch) -> 11.17 ms per 98 304 frames with position / score written before the sample loop and
// the sample counts in LDS (12 B of scratch left); with two parts the same compilation for 128 VGPRs was SLOWER (15.8 vs 14.8 ms at the
// time: spills without the occupancy).
#ifndef TH_PARTS
#define TH_PARTS 3
#endif
#define TH_JOINTS ((SE3D_NKP + TH_PARTS - 1) / TH_PARTS)   // joints per workgroup
struct TriHypLds {
  float x[TH_JOINTS][TH_MAXV], y[TH_JOINTS][TH_MAXV], cf[TH_JOINTS][TH_MAXV];  // normalised point and confidence per (joint, attached view); cf < 0: unusable
  float ca[TH_JOINTS][TH_MAXV], cb[TH_JOINTS][TH_MAXV], cc[TH_JOINTS][TH_MAXV];  // Cholesky factor of the 2x2 keypoint covariance
  float rows[TH_JOINTS][TH_MAXV][8];                   // DLT rows r0, r1 of every (joint, view); at the very end: 64 x 6 partial covariance sums
  float re[TH_JOINTS][TH_MAXV];                        // reprojection error of the weighted solution per (joint, view)
  double G[TH_JOINTS][20];                             // unweighted Gram (10), weighted Gram (10)
  float X[TH_JOINTS], Y[TH_JOINTS], Z[TH_JOINTS];
  int cam[TH_MAXV], det[TH_MAXV];
  int n[TH_JOINTS];
  int cnt[TH_JOINTS];                                  // sigma samples of the joint (0: no covariance from this kernel)
  int bad[TH_JOINTS];                                  // a usable view's 2x2 covariance is not SPD: the joint's 3-D covariance is NaN
  unsigned char vidx[TH_JOINTS][TH_MAXV];              // usable views of a joint in ascending camera order
};

// Round 4: five wavefronts per SIMD (96 VGPRs, 16 B of scratch per lane outside the sample loop).  With the Jacobi fallback out of the sample
// loop (a sample that does not settle sends the joint to the second pass) the kernel needs 101 registers; at 96 it runs 7.58 instead of 7.71 ms
// per 98 304 frames (profiles/r4/README.md).
#ifndef TH_WPE
#define TH_WPE 5
#endif
#ifdef SE3D_LM_STAMPS
__device__ unsigned long long g_th_stamps[16];
#define TH_STAMP_INIT unsigned long long th_prev_ = __builtin_readcyclecounter();
#define TH_STAMP(idx) { const unsigned long long now_ = __builtin_readcyclecounter(); if (threadIdx.x == 0 && ((blockIdx.x + 7 * blockIdx.y) & 63) == 0) atomicAdd(&g_th_stamps[idx], now_ - th_prev_); th_prev_ = now_; }
extern "C" void se3d_diag_th_stamps(unsigned long long *out16, int reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_th_stamps), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_th_stamps), z, sizeof(z)); }
}
#else
#define TH_STAMP_INIT
#define TH_STAMP(idx)
#endif
__global__ __launch_bounds__(64, TH_WPE) void se3d_k_triangulate_hyp(Se3dDev D) {
  __shared__ TriHypLds S;   // one wavefront per workgroup
  __shared__ float sRt[12 * TH_MAXV];
  __shared__ float sK[4 * TH_MAXV];
  const int lane = threadIdx.x;
  const int f = blockIdx.y, h = blockIdx.x / TH_PARTS, part_ = blockIdx.x - h * TH_PARTS;
  const int jbase = part_ * TH_JOINTS, JN = min(TH_JOINTS, SE3D_NKP - jbase);  // joints jbase .. jbase + JN - 1
  const int N = D.d.N, H = D.d.H;
  TH_STAMP_INIT
  // Everything whose address is known at entry is requested at once (hypothesis count, view count, association row, camera matrices): one
  // memory round trip, then the exits.  (h < H: the association row of an unused slot is readable.)
  const int nh_ = D.nhyp[f], nvw_ = D.nviews[f * H + h];
  const int ka = (lane < N) ? D.assoc[((size_t)f * H + h) * N + lane] : -1;
  float rt_[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) rt_[q] = (lane + 64 * q < 12 * N) ? D.Rt[lane + 64 * q] : 0.f;
  const float kin_ = (lane < 4 * N) ? D.K[lane] : 0.f;   // N <= 16: one intrinsic per lane
  if (h >= nh_) return;
  if (nvw_ < 2) return;
  if (D.d.shard_world > 1 && h % D.d.shard_world != D.d.shard_rank) return;  // hypothesis shard: another rank's hypothesis
#pragma unroll
  for (int q = 0; q < 3; ++q)
    if (lane + 64 * q < 12 * N) sRt[lane + 64 * q] = rt_[q];
  sK[lane] = kin_;
  TH_STAMP(0)  // entry loads
  // ---- attached cameras of the hypothesis, ascending; the camera's lane resolves the filtered slot to the raw detection index once ----
  int nv;
  {
    const unsigned long long bl = __ballot(ka >= 0);
    nv = __popcll(bl);
    if (ka >= 0) {
      const size_t cbase = ((size_t)(f * N + lane)) * D.d.P;
      const int pos = __popcll(bl & ((1ull << lane) - 1ull));
      S.cam[pos] = lane; S.det[pos] = (int)D.orig[cbase + ka];
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (part_ == 0) se3d_zero_unused_slots(D.hypP[(size_t)f * H + h], D.d.pose_method, lane);
  TH_STAMP(1)  // association row, unused slots
  // ---- points: lane <-> (joint, view slot), 16 view slots per joint and four joints per trip, so that a joint's views sit in one 16-lane
  // group: the list of usable views (ascending), their number and the bad-covariance flag come from two ballots instead of a serial
  // scan of the views by one lane per joint.  Row N on the fly from the raw payload (expressions of se3d_load_norm). ----
  // (joint, view) pairs packed densely over the lanes: item = lane + 64 trip, joint = item / nv, view = item % nv — one trip for up to
  // 64 / JN views per joint (10 or 12), two for more.  The usable-view lists come from the trips' ballots, read as one 128-bit mask:
  // the rank of a view among its joint's usable ones is a population count below its bit, wherever the joint's bits lie.
  static_assert(TH_MAXV == 16 && TH_JOINTS * TH_MAXV <= 128, "two trips of 64 (joint, view) items at most");
  const int nitems = JN * nv, ntrips = nitems > 64 ? 2 : 1;
  int ij[2], iv[2], ic[2];
  se3d_keypoint2d raw2[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {   // the payload of both trips is requested before either is worked on
    const int item = lane + 64 * t;
    ij[t] = item / nv; iv[t] = item - ij[t] * nv;
    raw2[t].x = raw2[t].y = 0.f; raw2[t].score = -1.f; raw2[t].cov[0] = raw2[t].cov[1] = raw2[t].cov[2] = 0.f;
    ic[t] = 0;
    if (item < nitems) {
      ic[t] = S.cam[iv[t]];
      raw2[t] = D.raw[((size_t)(f * N + ic[t])) * D.d.P + S.det[iv[t]]].keypoints[jbase + ij[t]];
    }
  }
  unsigned long long okm[2] = {0ull, 0ull}, badm[2] = {0ull, 0ull}, hicf[2] = {0ull, 0ull};   // hicf: a usable view with a confidence above 1
  bool okl[2] = {false, false};
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (t >= ntrips) break;   // uniform
    const int j = ij[t], v = iv[t], c = ic[t];
    const bool in = lane + 64 * t < nitems;
    const se3d_keypoint2d r = raw2[t];
    bool ok = false, badv = false;
    if (in) {
      float r0[4] = {0.f, 0.f, 0.f, 0.f}, r1[4] = {0.f, 0.f, 0.f, 0.f};
      float px = 0.f, py = 0.f;
      ok = r.score >= TRI_THRESH;   // S3D:724-732
      if (ok) {
        const float fx = sK[c * 4 + 0], fy = sK[c * 4 + 1], cx = sK[c * 4 + 2], cy = sK[c * 4 + 3];
        px = (r.x - cx) / fx; py = (r.y - cy) / fy;
        float k00, k01, k11, l11, l21, l22;
        se3d_norm_cov2(r.cov, fx, fy, k00, k01, k11);
        se3d_chol2(k00, k01, k11, l11, l21, l22);
        S.ca[j][v] = l11; S.cb[j][v] = l21; S.cc[j][v] = l22;
        badv = !(isfinite(l11) && isfinite(l21) && isfinite(l22));
        se3d_dlt_rows(sRt + 12 * c, px, py, r0, r1);
      }
      S.x[j][v] = px; S.y[j][v] = py; S.cf[j][v] = ok ? r.score : -1.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) { S.rows[j][v][k] = r0[k]; S.rows[j][v][4 + k] = r1[k]; }   // zero rows for an unusable view: the Gram loop adds them
    }
    okm[t] = __ballot(ok); badm[t] = __ballot(badv);
    hicf[t] = __ballot(ok && r.score > 1.0f);
    okl[t] = ok;
  }
  {
    // bits [b, b + nv) of the 128-bit mask (m1 : m0), nv <= 16
    auto seg = [&](unsigned long long m0, unsigned long long m1, int b2) -> unsigned {
      const unsigned long long x = b2 < 64 ? ((m0 >> b2) | (b2 ? (m1 << (64 - b2)) : 0ull)) : (m1 >> (b2 - 64));
      return (unsigned)x & ((1u << nv) - 1u);
    };
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t >= ntrips) break;
      const int j = ij[t], v = iv[t];
      if (lane + 64 * t < nitems) {
        const unsigned su = seg(okm[0], okm[1], j * nv);
        if (okl[t]) S.vidx[j][__popc(su & ((1u << v) - 1u))] = (unsigned char)v;
        if (v == 0) { S.n[j] = __popc(su); S.bad[j] = seg(badm[0], badm[1], j * nv) ? 1 : 0; }
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  TH_STAMP(2)  // points + rows + view lists
  // ---- Gram matrices: lane <-> (joint, entry), unweighted and weighted entry together; accumulation over the view slots in ascending order
  // like the per-joint kernel (an unusable view's rows are zero: adding them changes nothing).  The sum of confidences is formed by the score
  // loop below (same terms, same order). ----
  for (int it = lane; it < JN * 10; it += 64) {   // one trip for up to six joints
    const int j = it / 10, e = it - 10 * j;
    {
      double au = 0.0, aw = 0.0;
      if (S.n[j] >= 2) {
        const int gi = (int)((0x3221110000ull >> (4 * e)) & 0xFull), gj = (int)((0x3323213210ull >> (4 * e)) & 0xFull);
#pragma unroll 4
        for (int v = 0; v < nv; ++v) {
          const float cf = S.cf[j][v];
          const float a0 = S.rows[j][v][gi], b0 = S.rows[j][v][gj], a1 = S.rows[j][v][4 + gi], b1 = S.rows[j][v][4 + gj];
          au = __builtin_fma((double)a0, (double)b0, au);
          au = __builtin_fma((double)a1, (double)b1, au);
          const float wa0 = a0 * cf, wb0 = b0 * cf, wa1 = a1 * cf, wb1 = b1 * cf;
          aw = __builtin_fma((double)wa0, (double)wb0, aw);
          aw = __builtin_fma((double)wa1, (double)wb1, aw);
        }
      }
      S.G[j][e] = au; S.G[j][10 + e] = aw;
    }
  }
  __builtin_amdgcn_wave_barrier();
  TH_STAMP(4)  // Gram
  // ---- weighted solve, reprojection error and score: lane <-> joint ----
  const int nj = (lane < JN) ? S.n[lane] : 0;
  const bool active = nj >= 2;
  bool slow = false;
  float X = 0.f, Y = 0.f, Z = 0.f, avg_score = 0.f;
  bool unsettled = false;   // the inverse iteration of the weighted system fell back to the Jacobi solver: second pass
  bool level2 = false;      // the cheap sufficient condition for "well conditioned" failed: the full test below decides
  if (active) {
    double gw_[10], pw[3];
#pragma unroll
    for (int k = 0; k < 10; ++k) gw_[k] = S.G[lane][10 + k];
    const double e4start[4] = {0.0, 0.0, 0.0, 1.0};
    double piv_[4];
    if (!se3d_null_invit_point<2, 12>(gw_, e4start, pw, SE3D_MAIN_PTOL, piv_)) {
      double e4[4];
      se3d_eig4_smallest(gw_, e4);
      const double iw = se3d_rcp64<2>(e4[3]);
      pw[0] = e4[0] * iw; pw[1] = e4[1] * iw; pw[2] = e4[2] * iw;
      unsettled = true;
    } else {
      level2 = !se3d_cond_level1_ok(gw_, piv_, pw, nj);
    }
    X = (float)pw[0]; Y = (float)pw[1]; Z = (float)pw[2];
    S.X[lane] = X; S.Y[lane] = Y; S.Z[lane] = Z;   // also the start (X, Y, Z, 1) of the samples' solves
  }
  __builtin_amdgcn_wave_barrier();
  // Conditioning (round 4): joints that failed the level-1 test above (or carry a confidence above 1, which that test does not cover) take the
  // full test of the weighted and of the unweighted system, lane <-> (system, joint): 2 JN <= 12 lanes at once; skipped by the wavefront when no
  // joint needs it.  The published float32 point stands in for the fp64 solution (the Rayleigh quotient only grows).
  unsigned long long illmask = 0ull;
  {
    unsigned long long l2mask = __ballot(level2);
    if ((hicf[0] | hicf[1]) != 0ull) l2mask = (1ull << JN) - 1ull;
    if (l2mask) {
      bool ill = false;
      if (lane < 2 * JN) {
        const int which = lane >= JN ? 1 : 0, jq = lane - which * JN;
        if (S.n[jq] >= 2 && ((l2mask >> jq) & 1ull)) {
          float gq_[10];
#pragma unroll
          for (int k = 0; k < 10; ++k) gq_[k] = (float)S.G[jq][10 * which + k];
#ifndef SE3D_ABLATE_NO_KAPPA   // ablation build only (tools/profile_r4.sh ablate): what the test itself costs
          ill = se3d_kappa_flag(gq_, S.X[jq], S.Y[jq], S.Z[jq]);
#endif
        }
      }
      illmask = __ballot(ill);
    }
  }
  TH_STAMP(5)  // weighted solve
#pragma unroll
  for (int t = 0; t < 2; ++t) {  // confidence-weighted reprojection errors on the same (joint, view) lanes; 0 for a view that is not usable
    if (t >= ntrips) break;
    const int j = ij[t], v = iv[t];
    if (lane + 64 * t < nitems) {
      const float cf = S.cf[j][v];
      float ce = 0.f;
      if (S.n[j] >= 2 && cf >= 0.f) ce = cf * se3d_reproj1(sRt + 12 * ic[t], S.X[j], S.Y[j], S.Z[j], S.x[j][v], S.y[j][v]);
      S.re[j][v] = ce;
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (active) {
    double se = 0.0, sc = 0.0;
    float scf = 0.f;
#pragma unroll 4
    for (int v = 0; v < nv; ++v) {  // summed in view order like the per-joint kernel (the terms of the views that are not usable are zero)
      const float cfv = fmaxf(S.cf[lane][v], 0.f);
      se += (double)S.re[lane][v];
      sc += (double)cfv;
      scf += cfv;   // the reference's own float sum over the usable views in view order (S3D:730); + 0.f is the identity
    }
    // quotient by reciprocal (hardware seed + two Newton steps, <= 1 ulp of fp64 from the IEEE quotient) instead of an fp64 division
    // sequence: the error only meets thresholds.  The score is the reference's float expression (S3D:738) — a joint that stays on this path
    // has err <= g_reproj_error_max_acceptable, so S3D:840-841 leave it alone: bit-identical scores.
    const double err = se * se3d_rcp64(sc);
    avg_score = scf / (float)nj;
    // second pass (round 4): outlier rejection (S3D:748-838) and everything near / above its threshold (the score carries err, S3D:840-844),
    // ill-conditioned systems, unsettled solves
    slow = !(err <= REPROJ_MAX * (1.0 - SE3D_EXACT_ERR_GUARD)) || unsettled || ((illmask >> lane) & 1ull) || ((illmask >> (lane + JN)) & 1ull);
  }
  const unsigned long long slowmask = __ballot(slow);
  TH_STAMP(6)  // reprojection errors + score
  // Position, score and the default-constructed slots are written HERE, before the sample loop: nothing below needs X, Y, Z, the score or the
  // flags in registers any more (the loop re-reads the centre from LDS), which is what lets the kernel fit 128 VGPRs without spilling them
  // across the loop (56 B of scratch per lane otherwise: 5.5 GB of HBM traffic per 98 304-frame launch in the PMC passes).
  {
    se3d_person3d &person0 = D.hypP[(size_t)f * H + h];
    const int pm_ = D.d.pose_method;
  if (lane < JN && !slow) {  // second-pass joints are written (all fields) by se3d_k_triangulate_exact
    se3d_keypoint3d *kp = &person0.keypoints[c_kp2fusion[pm_][jbase + lane]];
    if (active) {
      kp->joint[0] = (double)X; kp->joint[1] = (double)Y; kp->joint[2] = (double)Z;
      kp->score = avg_score; kp->_pad = 0.f;
    } else {  // < 2 usable views (S3D:734-736): default-constructed slot
      kp->joint[0] = kp->joint[1] = kp->joint[2] = 0.0;
      kp->score = 0.f; kp->_pad = 0.f;
      for (int e = 0; e < 6; ++e) kp->cov[e] = 0.0;
    }
  }
  // ---- joints of the second pass (se3d_k_triangulate_exact) go to the list ----
  if (slowmask) {   // uniform; rare
    uint32_t at0 = 0u;
    if (lane == 0) at0 = atomicAdd(&D.tri_cnt[0], (uint32_t)__popcll(slowmask));
    at0 = (uint32_t)__shfl((int)at0, 0);
    if (slow) D.tri_list[at0 + __popcll(slowmask & ((1ull << lane) - 1ull))] = ((uint32_t)(f * H + h) << 5) | (uint32_t)(jbase + lane);
  }
  }
  // ---- sigma points (S3D:471-523): lanes are tied to joints (64 / JN lanes each) and stride over the joint's 4n + 1 samples, so the
  // joint's Gram matrix, start vector and centre stay in registers and the weighted outer products of the deviations are
  // accumulated in registers as the samples are solved (no sample buffer: 7 KB of LDS less, the DLT rows of the views stay
  // resident instead of being rebuilt for every sample); the partial sums of a joint's lanes are added in lane order.
  if (lane < JN) S.cnt[lane] = (active && !slow) ? 4 * nj + 1 : 0;
  TH_STAMP(7)  // position / score writes
  __builtin_amdgcn_wave_barrier();
  {
    const int LPJ = 64 / JN;                       // 10 lanes per joint (6 joints) or 12 (5 joints)
    const int jl = lane / LPJ, j = min(jl, JN - 1), s0 = lane - LPJ * jl;
    const int cntj = jl < JN ? S.cnt[j] : 0;
    const int n = S.n[j];
    const float Xj = S.X[j], Yj = S.Y[j], Zj = S.Z[j];
    const float bsc = sqrtf((float)(2 * n) + 0.5f);
    const float wgt = 1.0f / (2.0f * ((float)(2 * n) + 0.5f));  // all weights equal because 2*kappa == 1
    bool sample_bad = false;   // a sample's inverse iteration did not settle: the joint goes to the second pass
    const double vstart[4] = {(double)Xj, (double)Yj, (double)Zj, 1.0};   // the iteration is scale-free: the published point is its start
    double gj[10], cv[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 10; ++k) gj[k] = S.G[j][k];
    // class A (centre + 2n diagonal samples: both rows rebuilt) first, then class B (2n y-only samples: r1 alone); the lanes of all
    // joints are in the same class except around the switch
    const int cntA = cntj > 0 ? 2 * n + 1 : 0, cntB = cntj > 0 ? 2 * n : 0;
    for (int cls = 0; cls < 2; ++cls) {
      for (int s = s0; s < (cls == 0 ? cntA : cntB); s += LPJ) {
        double gq[10];
#pragma unroll
        for (int k = 0; k < 10; ++k) gq[k] = gj[k];
        const int q = cls == 0 ? s - 1 : s;            // class A: s = 0 is the centre
        if (q >= 0) {
          const int c = q >> 1, v = S.vidx[j][c];
          float o0[4], o1[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) { o0[k] = S.rows[j][v][k]; o1[k] = S.rows[j][v][4 + k]; }
          se3d_sigma_gram(gq, sRt + 12 * S.cam[v], S.x[j][v], S.y[j][v], S.ca[j][v], S.cb[j][v], S.cc[j][v], bsc, o0, o1, (cls == 0 ? 1 : 3) + (q & 1));
        }
        double pt[3];
        sample_bad |= !se3d_null_invit_point(gq, vstart, pt);   // (no Jacobi fallback in the loop: a sample that does not settle sends the joint to the second pass)
        const float sx = (float)pt[0] - Xj, sy = (float)pt[1] - Yj, sz = (float)pt[2] - Zj;
        const double wx = (double)(sx * wgt), wy = (double)(sy * wgt), wz = (double)(sz * wgt);
        cv[0] += wx * sx; cv[1] += wx * sy; cv[2] += wx * sz; cv[3] += wy * sy; cv[4] += wy * sz; cv[5] += wz * sz;
      }
    }
    TH_STAMP(8)  // sigma samples
    const unsigned long long sbad = __ballot(sample_bad);   // lane <-> (joint jl, sample stride): bits [LPJ jl, LPJ (jl + 1)) belong to joint jl
    __builtin_amdgcn_wave_barrier();   // all samples done: the rows are dead, their LDS takes the partial sums
    // (rows, and behind them re / vs / G, are all dead once the samples are done)
    static_assert(offsetof(TriHypLds, X) - offsetof(TriHypLds, rows) >= 64 * 6 * sizeof(double) && offsetof(TriHypLds, rows) % 8 == 0, "partial sums alias the rows");
    double (*part)[6] = reinterpret_cast<double (*)[6]>(&S.rows[0][0][0]);
#pragma unroll
    for (int e = 0; e < 6; ++e) part[lane][e] = cv[e];
    __builtin_amdgcn_wave_barrier();
    static_assert(TH_JOINTS * 6 <= 64, "one trip over the (joint, entry) lanes");
    {
      const int it = lane, jj = it / 6, e = it - jj * 6;
      bool wide = false;   // a variance above SE3D_EXACT_VAR_MAX or a sample whose solve did not settle: the joint goes to the second pass
      if (it < JN * 6 && S.cnt[jj] != 0) {
        double acc = 0.0;
        for (int q = 0; q < LPJ; ++q) acc += part[jj * LPJ + q][e];
        if (S.bad[jj]) acc = (double)__builtin_nanf("");
        else wide = ((e == 0 || e == 3 || e == 5) && !(acc < SE3D_EXACT_VAR_MAX)) || (e == 0 && ((sbad >> (LPJ * jj)) & ((1ull << LPJ) - 1ull)) != 0ull);
        D.hypP[(size_t)blockIdx.y * D.d.H + blockIdx.x / TH_PARTS].keypoints[c_kp2fusion[D.d.pose_method][jbase + jj]].cov[e] = (double)(float)acc;   // (address re-formed here: not carried across the loop)
      }
      const unsigned long long wmask = __ballot(wide);
      if (wmask) {   // uniform; rare
        bool mine = false;
        if (it < JN * 6 && e == 0) mine = ((wmask >> (6 * jj)) & 0x29ull) != 0ull;   // entries 0, 3, 5 of joint jj
        const unsigned long long jm = __ballot(mine);
        uint32_t at0 = 0u;
        if (lane == 0) at0 = atomicAdd(&D.tri_cnt[0], (uint32_t)__popcll(jm));
        at0 = (uint32_t)__shfl((int)at0, 0);
        if (mine) D.tri_list[at0 + __popcll(jm & ((1ull << lane) - 1ull))] = ((uint32_t)(blockIdx.y * D.d.H + blockIdx.x / TH_PARTS) << 5) | (uint32_t)(jbase + jj);
      }
    }
    TH_STAMP(9)  // covariance sums + writes
#ifdef SE3D_LM_STAMPS
    if (threadIdx.x == 0 && ((blockIdx.x + 7 * blockIdx.y) & 63) == 0) atomicAdd(&g_th_stamps[15], 1ull);
#endif
  }
}

// =============================================================================================================
// Second pass of the triangulation (round 4): rows T, P, O, U of the listed joints with the oracle's restatement of the reference's float32 arithmetic.
// One wavefront per joint, one DLT solve per lane (se3d_exact32.h):
//   phase A  lane 0 solves the confidence-weighted system of all n usable views (S3D:746), lanes 1 .. n its leave-one-view-out subsets
//            (S3D:793-838; with three views the subset left after the epipolar test of S3D:748-792 is one of them) — speculatively: the lanes
//            would idle otherwise; every lane takes the reprojection error of its own point over its own views (S3D:425-438);
//   decision the reference's scans in their order (first strict minimum wins), score = mean confidence of the kept views in float (S3D:730-742,
//            760-762), down-weighting (S3D:840-844);
//   phase B  the 4n' + 1 sigma samples of the kept views, one unweighted solve per lane (S3D:471-518); the weighted outer products are summed
//            in sample order in float (S3D:521-522).
// Every expression and every summation order is the one the CPU restatement of the reference uses (tests/: bit-identical positions, scores and
// raw covariances on the listed joints).  LDS per wavefront: camera matrices, the view list, the views' normalised rows, the centred samples.
// =============================================================================================================
struct X32Views {
  float *x, *y, *cf, *k00, *k01, *k11;   // normalised point, confidence, normalised 2x2 covariance (a00, a01 = a10, a11) per usable view
  float *rows;                            // [n][8] DLT rows r0, r1 of the view, normalised by division (S3D:446-449), unweighted
  int *cam;
};
// LDS of a wavefront of the second pass: view list and rows; on small rigs also the camera matrices and the centred samples.  Large rigs
// (N > SE3D_EXACT_SMALL_N) read the camera matrices from global memory and keep the centred samples behind the lanes' matrices in the
// workspace: 15 KB instead of 40 KB at 256 cameras, ten wavefronts per CU instead of four.
#define SE3D_EXACT_SMALL_N 64
__host__ __device__ inline size_t se3d_exact_lds_bytes(int N) {
  const size_t small = N <= SE3D_EXACT_SMALL_N ? (size_t)12 * N + 3 * (4 * N + 1) : 0;
  return (((small + 6 * N + 8 * N) * sizeof(float) + (size_t)N * sizeof(int)) + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t se3d_exact_ws_floats(int N) {   // per wavefront: 64 lane-interleaved 2N x 4 matrices (+ the centred samples on large rigs)
  return (size_t)64 * 8 * N + (N <= SE3D_EXACT_SMALL_N ? 0 : (((size_t)3 * (4 * N + 1) + 63) & ~(size_t)63));
}

// one triangulate() call of the reference on this lane: views 0 .. n-1 without `skip`, view `mv` (>= 0) taking the point (mx, my)
template <typename WsPtrT>
__device__ __forceinline__ void x32_triangulate(const X32Views &T, const float *__restrict__ sRt, int n, bool weighted, int skip, int mv, float mx, float my,
                                                const X32MatT<WsPtrT> &A, float X[3]) {
  int r = 0;
  float amax = 0.f;   // largest |entry| of the system (JacobiSVD's scale), tracked while the rows are written
  for (int v = 0; v < n; ++v) {
    if (v == skip) continue;
    float r0[4], r1[4];
    if (v == mv) {
      x32_dlt_rows(sRt + 12 * T.cam[v], mx, my, r0, r1);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) { r0[k] = T.rows[8 * v + k]; r1[k] = T.rows[8 * v + 4 + k]; }
    }
    if (weighted) {   // A.row(2i) *= conf (S3D:450-453)
      const float cf = T.cf[v];
#pragma unroll
      for (int k = 0; k < 4; ++k) { r0[k] = r0[k] * cf; r1[k] = r1[k] * cf; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { A.set(r, k, r0[k]); A.set(r + 1, k, r1[k]); amax = fmaxf(amax, fmaxf(fabsf(r0[k]), fabsf(r1[k]))); }
    r += 2;
  }
  const float4 v4 = x32_null_vector(r, A.p, A.stride, A.ld, amax);
  X[0] = v4.x / v4.w; X[1] = v4.y / v4.w; X[2] = v4.z / v4.w;   // hnormalized()
}
// calcReprojectionError (S3D:425-438) of X over the views 0 .. n-1 without `skip`
__device__ __forceinline__ double x32_reproj_error(const X32Views &T, const float *__restrict__ sRt, int n, int skip, const float X[3]) {
  double averageError = 0., norm = 0.;
  for (int v = 0; v < n; ++v) {
    if (v == skip) continue;
    const float *M = sRt + 12 * T.cam[v];
    const float hx = (M[0] * X[0] + (M[1] * X[1] + M[2] * X[2])) + M[3];
    const float hy = (M[4] * X[0] + (M[5] * X[1] + M[6] * X[2])) + M[7];
    const float hz = (M[8] * X[0] + (M[9] * X[1] + M[10] * X[2])) + M[11];
    const float ix = hx / hz, iy = hy / hz;
    const float dx = ix - T.x[v], dy = iy - T.y[v];
    const float error = sqrtf(dx * dx + dy * dy);
    averageError += (double)(T.cf[v] * error);
    norm += (double)T.cf[v];
  }
  return averageError / norm;
}

// =============================================================================================================
// Round 6: FOUR listed joints per wavefront on small rigs (N <= 16).  The one-joint kernel below keeps n + 2 of its 64 lanes busy in phase A — the
// weighted system and its n leave-one-view-out subsets, ~11 lanes at the benchmark's 9 views per joint — and the counters of the rough-scene pass
// (profiles/r6/pmc_sq_16cam6p_rough_split.json) show the vector ALU 78 % busy at that lane utilisation: the pass is bound by wave-INSTRUCTIONS, so the
// lever is lanes per instruction.  Here a 16-lane group carries one joint (lane 0 the weighted system, lanes 1 .. n its subsets: n <= 15), four
// groups share every instruction of the solve; the decisions of S3D:748-844 are taken per group with shuffles inside the group; phase B runs the fp64
// sample solves (the covariance contract of the fast path) 16 samples per group and round.  What this kernel does NOT do it hands to the one-joint
// kernel by setting bit 30 of the list entry: a quad with a joint seen by all 16 cameras, and every joint whose samples need float32 arithmetic (ill
// conditioned kept views, wide covariance, unsettled sample: ~8 % of the listed joints of the rough scenes).  Same device functions, same expressions
// and orders per joint as the one-joint kernel: positions, scores and decisions carry the oracle's bits (tests/test_gpu_exact32.py).
// =============================================================================================================
#define X4_GL 16                       // lanes per joint
#define X4_G (64 / X4_GL)              // joints per wavefront
#define SE3D_EXACT4_MAX_N 16           // rigs this kernel is launched for
__host__ __device__ inline size_t se3d_exact4_lds_bytes(int N) {   // camera matrices + per group the view list (6 N floats), its rows (8 N) and cameras (N ints)
  return (((size_t)12 * N + (size_t)X4_G * (14 * N)) * sizeof(float) + (size_t)X4_G * N * sizeof(int) + 15) & ~(size_t)15;
}
__device__ __forceinline__ double x4_group_sum(double v) {
#pragma unroll
  for (int o = X4_GL / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double x4_shfl_f64(double v, int src) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__shfl((int)(unsigned)u, src), hi = (unsigned)__shfl((int)(unsigned)(u >> 32), src);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__global__ __launch_bounds__(64, 3) void se3d_k_triangulate_exact4(Se3dDev D) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x, grp = lane / X4_GL, gl = lane % X4_GL, base = grp * X4_GL;
  const int N = D.d.N, H = D.d.H;
  const uint32_t cnt = D.tri_cnt[0];
  const uint32_t nquads = (cnt + X4_G - 1) / X4_G;
  if (blockIdx.x >= nquads) return;
  float *sRt = reinterpret_cast<float *>(smem);
  for (int i = lane; i < 12 * N; i += 64) sRt[i] = D.Rt[i];
  float *gbase = sRt + 12 * N + (size_t)grp * 14 * N;
  X32Views T;
  T.x = gbase; T.y = T.x + N; T.cf = T.y + N; T.k00 = T.cf + N; T.k01 = T.k00 + N; T.k11 = T.k01 + N;
  T.rows = T.k11 + N;
  T.cam = reinterpret_cast<int *>(sRt + 12 * N + (size_t)X4_G * 14 * N) + grp * N;
  const X32Mat A{D.tri_ws + (size_t)blockIdx.x * se3d_exact_ws_floats(N) + lane, 64, 2 * N};
  __syncthreads();
  for (uint32_t quad = blockIdx.x; quad < nquads; quad += gridDim.x) {
    const uint32_t it = quad * X4_G + grp;
    const bool have = it < cnt;
    const uint32_t en = have ? (D.tri_list[it] & 0x3fffffffu) : 0u;
    const int e = (int)(en >> 5), j = (int)(en & 31u), f = e / H, h = e - f * H;
    // ---- usable views of the group's joint (conf >= 0.30, S3D:724-732), ascending camera order: lane gl <-> camera gl (N <= 16) ----
    bool ok = false;
    se3d_keypoint2d r;
    r.x = r.y = 0.f; r.score = -1.f; r.cov[0] = r.cov[1] = r.cov[2] = 0.f;
    if (have && gl < N) {
      const int k = D.assoc[((size_t)f * H + h) * N + gl];
      if (k >= 0) {
        const size_t cb = ((size_t)(f * N + gl)) * D.d.P;
        r = D.raw[cb + D.orig[cb + k]].keypoints[j];
        ok = r.score >= TRI_THRESH;
      }
    }
    const unsigned bl = (unsigned)((__ballot(ok) >> base) & 0xffffull);
    int n = __popc(bl);
    if (ok) {
      const int pos = __popc(bl & ((1u << gl) - 1u));
      const float fx = D.K[gl * 4 + 0], fy = D.K[gl * 4 + 1], cx = D.K[gl * 4 + 2], cy = D.K[gl * 4 + 3];
      T.x[pos] = (r.x - cx) / fx; T.y[pos] = (r.y - cy) / fy; T.cf[pos] = r.score;
      T.k00[pos] = r.cov[0] / (fx * fx); T.k01[pos] = r.cov[1] / (fx * fy); T.k11[pos] = r.cov[2] / (fy * fy);   // S3D:324-326
      T.cam[pos] = gl;
    }
    // a quad with a joint that does not fit a group (n + 1 > 16 lanes) or cannot be triangulated goes to the one-joint kernel as a whole (uniform)
    if (__ballot(have && (n > X4_GL - 1 || n < 2)) != 0ull) {
      if (have && gl == 0) D.tri_list[it] = en | 0x40000000u;
      __syncthreads();
      continue;
    }
    __syncthreads();
    if (have && gl < n) {
      float r0[4], r1[4];
      x32_dlt_rows(sRt + 12 * T.cam[gl], T.x[gl], T.y[gl], r0, r1);
#pragma unroll
      for (int k = 0; k < 4; ++k) { T.rows[8 * gl + k] = r0[k]; T.rows[8 * gl + 4 + k] = r1[k]; }
    }
    __syncthreads();
    // ---- phase A: the weighted system (group lane 0) and, with three or more views, its leave-one-view-out subsets (group lanes 1 .. n) ----
    float Xl[3] = {0.f, 0.f, 0.f};
    double el = 1e300;
    if (have && (gl == 0 || (n >= 3 && gl <= n))) {
      x32_triangulate(T, sRt, n, true, gl - 1, -1, 0.f, 0.f, A, Xl);
      el = x32_reproj_error(T, sRt, n, gl - 1, Xl);
    }
    float X = __shfl(Xl[0], base), Y = __shfl(Xl[1], base), Z = __shfl(Xl[2], base);
    double err = x4_shfl_f64(el, base);
    float avg_score = 0.f;
    for (int v = 0; v < n; ++v) avg_score += T.cf[v];   // S3D:730 (every lane of the group forms the same sum)
    avg_score /= (float)(n > 0 ? n : 1);                // S3D:738
    int rm = -1;
    {   // S3D:748-792 (three views): the pair of remaining views with the smallest squared symmetric epipolar distance
      const bool three = have && err > REPROJ_MAX && n == 3;
      float dcand = 0.f;
      if (three && gl < 3) {
        const int i0 = (gl == 0) ? 1 : 0, i1 = (gl == 2) ? 1 : 2;
        const float p1x = T.x[i0], p1y = T.y[i0], p2x = T.x[i1], p2y = T.y[i1];
        const float *F = D.F + 9 * (size_t)se3d_pair_index(T.cam[i0], T.cam[i1], N);
        const float l1x = F[0] * p1x + (F[1] * p1y + F[2]);
        const float l1y = F[3] * p1x + (F[4] * p1y + F[5]);
        const float l1z = F[6] * p1x + (F[7] * p1y + F[8]);
        const float l2x = F[0] * p2x + (F[3] * p2y + F[6]);
        const float l2y = F[1] * p2x + (F[4] * p2y + F[7]);
        const float l2z = F[2] * p2x + (F[5] * p2y + F[8]);
        const float n1 = p2x * l1x + (p2y * l1y + l1z);
        const float n2 = p1x * l2x + (p1y * l2y + l2z);
        dcand = n1 * n1 / (l1x * l1x + l1y * l1y) + n2 * n2 / (l2x * l2x + l2y * l2y);
      }
      float best = (float)(err * err);
      int rm3 = -1;
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const float di = __shfl(dcand, base + i);
        if (di < best) { best = di; rm3 = i; }
      }
      // S3D:793-838 (four or more views): candidates in ascending view order, first strict minimum below 0.9 x err wins
      const bool loo = have && err > REPROJ_MAX && n >= 4;
      double m = (loo && gl >= 1 && gl <= n && el < 0.9 * err) ? el : 1e300;
      int mi = gl - 1;
#pragma unroll
      for (int o = X4_GL / 2; o > 0; o >>= 1) {
        const double om = x4_shfl_f64(m, lane ^ o);
        const int oi = __shfl_xor(mi, o);
        if (om < m || (om == m && oi < mi)) { m = om; mi = oi; }
      }
      if (three && rm3 != -1) rm = rm3;
      if (loo && m < err) rm = mi;
      // (the shuffles below are executed by every lane; groups without a removal read their own lane 0 + (-1 + 1))
      const int src = base + (rm != -1 ? rm + 1 : 0);
      const float Xn = __shfl(Xl[0], src), Yn = __shfl(Xl[1], src), Zn = __shfl(Xl[2], src);
      const double en_ = x4_shfl_f64(el, src);
      if (rm != -1) {
        X = Xn; Y = Yn; Z = Zn;
        err = en_;
        if (n == 3) {
          const int a = rm == 0 ? 1 : 0, b = rm == 2 ? 1 : 2;
          avg_score = (T.cf[a] + T.cf[b]) / 2.0f;
        } else {
          float tmpScore = 0.f;
          for (int v = 0; v < n; ++v)
            if (v != rm) tmpScore += T.cf[v];
          avg_score = tmpScore / (float)(n - 1);
        }
      }
    }
    __syncthreads();
    if (rm != -1 && gl == 0) {   // erase the view, order preserved (one lane per group)
      for (int v = rm; v + 1 < n; ++v) {
        T.x[v] = T.x[v + 1]; T.y[v] = T.y[v + 1]; T.cf[v] = T.cf[v + 1];
        T.k00[v] = T.k00[v + 1]; T.k01[v] = T.k01[v + 1]; T.k11[v] = T.k11[v + 1]; T.cam[v] = T.cam[v + 1];
        for (int k = 0; k < 8; ++k) T.rows[8 * v + k] = T.rows[8 * (v + 1) + k];
      }
    }
    if (rm != -1) n -= 1;
    __syncthreads();
    if (err > REPROJ_MAX) avg_score = (float)((double)avg_score * (REPROJ_MAX / err));  // S3D:840-841
    // ---- phase B: fp64 sample solves around the float32 point where the kept views are well conditioned (see the one-joint kernel), else bit 30 ----
    const int dim = 2 * n, nsamp = 4 * n + 1;
    const float bsc = sqrtf((float)dim + 0.5f);
    bool not_spd = false;
    if (have && gl < n) {
      const float l11 = sqrtf(T.k00[gl]), l21 = T.k01[gl] / l11, l22 = sqrtf(T.k11[gl] - l21 * l21);
      not_spd = !(isfinite(l11) && isfinite(l21) && isfinite(l22));
    }
    not_spd = ((__ballot(not_spd) >> base) & 0xffffull) != 0ull;
    double gu_[10], gw_[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) { gu_[k] = 0.0; gw_[k] = 0.0; }
    if (have && gl < n) {
      float r0[4], r1[4], w0[4], w1[4];
      const float cf = T.cf[gl];
#pragma unroll
      for (int k = 0; k < 4; ++k) { r0[k] = T.rows[8 * gl + k]; r1[k] = T.rows[8 * gl + 4 + k]; w0[k] = r0[k] * cf; w1[k] = r1[k] * cf; }
      se3d_gram_add(gu_, r0); se3d_gram_add(gu_, r1);
      se3d_gram_add(gw_, w0); se3d_gram_add(gw_, w1);
    }
    float gwf[10], guf[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) { gu_[k] = x4_group_sum(gu_[k]); gw_[k] = x4_group_sum(gw_[k]); guf[k] = (float)gu_[k]; gwf[k] = (float)gw_[k]; }
    bool cov64 = false;
    float cov6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool well = have && !not_spd && !(se3d_kappa_flag(gwf, X, Y, Z) || se3d_kappa_flag(guf, X, Y, Z));   // uniform per group
    {
      const float wgt = 1.0f / (2.0f * ((float)dim + 0.5f));
      double cv[6] = {0., 0., 0., 0., 0., 0.};
      bool sample_bad = false;
      const double vstart[4] = {(double)X, (double)Y, (double)Z, 1.0};
      for (int s0 = 0; s0 < 4 * (X4_GL - 1) + 1; s0 += X4_GL) {   // (<= 61 samples: four rounds; the trip count is uniform, the lanes are predicated)
        const int s = s0 + gl;
        if (well && s < nsamp) {
          double g[10];
#pragma unroll
          for (int k = 0; k < 10; ++k) g[k] = gu_[k];
          if (s > 0) {
            const bool clsB = s > 2 * n;
            const int q = clsB ? s - 2 * n - 1 : s - 1, c = q >> 1;
            float o0[4], o1[4], l11, l21, l22;
#pragma unroll
            for (int k = 0; k < 4; ++k) { o0[k] = T.rows[8 * c + k]; o1[k] = T.rows[8 * c + 4 + k]; }
            se3d_chol2(T.k00[c], T.k01[c], T.k11[c], l11, l21, l22);
            se3d_sigma_gram(g, sRt + 12 * T.cam[c], T.x[c], T.y[c], l11, l21, l22, bsc, o0, o1, (clsB ? 3 : 1) + (q & 1));
          }
          double pt[3];
          sample_bad |= !se3d_null_invit_point(g, vstart, pt);
          const float sx = (float)pt[0] - X, sy = (float)pt[1] - Y, sz = (float)pt[2] - Z;
          const double wx = (double)(sx * wgt), wy = (double)(sy * wgt), wz = (double)(sz * wgt);
          cv[0] += wx * sx; cv[1] += wx * sy; cv[2] += wx * sz; cv[3] += wy * sy; cv[4] += wy * sz; cv[5] += wz * sz;
        }
      }
#pragma unroll
      for (int k = 0; k < 6; ++k) cv[k] = x4_group_sum(cv[k]);
      const bool any_bad = ((__ballot(sample_bad) >> base) & 0xffffull) != 0ull;
      if (well && !any_bad && cv[0] < SE3D_EXACT_VAR_MAX && cv[3] < SE3D_EXACT_VAR_MAX && cv[5] < SE3D_EXACT_VAR_MAX) {
        cov64 = true;
#pragma unroll
        for (int k = 0; k < 6; ++k) cov6[k] = (float)cv[k];
      }
    }
    if (not_spd) {   // defined policy: all entries NaN, position and score unaffected
#pragma unroll
      for (int k = 0; k < 6; ++k) cov6[k] = __builtin_nanf("");
    }
    if (have && gl == 0) {
      if (cov64 || not_spd) {
        se3d_keypoint3d *kpo = &D.hypP[(size_t)f * H + h].keypoints[c_kp2fusion[D.d.pose_method][j]];
        kpo->joint[0] = (double)X; kpo->joint[1] = (double)Y; kpo->joint[2] = (double)Z;
        kpo->score = avg_score; kpo->_pad = 0.f;
#pragma unroll
        for (int k = 0; k < 6; ++k) kpo->cov[k] = (double)cov6[k];
        if (cov64) D.tri_list[it] = en | 0x80000000u;
      } else {
        D.tri_list[it] = en | 0x40000000u;   // float32 samples needed: the one-joint kernel redoes this joint
      }
    }
    __syncthreads();
  }
}

#ifndef X32_WPE
#define X32_WPE 3   // wavefronts per SIMD the second pass is compiled for: 164 VGPRs without scratch (4: 128 VGPRs + 80 B of scratch per lane)
#endif
__global__ __launch_bounds__(64, X32_WPE) void se3d_k_triangulate_exact(Se3dDev D, int cov32 /* options.exact_cov: float32 samples for every listed joint */, int only_marked /* the four-joint kernel ran first: only the entries it marked */) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int lane = threadIdx.x;
  const int N = D.d.N, H = D.d.H;
  const uint32_t cnt = D.tri_cnt[0];
  if (blockIdx.x == 0 && lane == 0) D.tri_cnt[1] += cnt;   // running total (se3d_get_exact_stats); launches of a context are serialised
  if (blockIdx.x >= cnt) return;
  const bool small_rig = N <= SE3D_EXACT_SMALL_N;
  float *const wsw = D.tri_ws + (size_t)blockIdx.x * se3d_exact_ws_floats(N);
  float *lds = reinterpret_cast<float *>(smem);
  const float *sRt = D.Rt;                           // camera matrices: LDS copy on small rigs, global memory (L1 / L2) on large ones
  float *cen = wsw + (size_t)64 * 8 * N;             // [4n + 1][3] centred samples: LDS on small rigs, workspace on large ones
  if (small_rig) {
    for (int i = lane; i < 12 * N; i += 64) lds[i] = D.Rt[i];
    sRt = lds;
    cen = lds + 12 * N;
    lds = cen + 3 * (4 * N + 1);
  }
  X32Views T;
  T.x = lds; T.y = T.x + N; T.cf = T.y + N; T.k00 = T.cf + N; T.k01 = T.k00 + N; T.k11 = T.k01 + N;
  T.rows = T.k11 + N;
  T.cam = reinterpret_cast<int *>(T.rows + 8 * N);
  // The lanes' rows x 4 matrices live in a lane-interleaved workspace in global memory (L2-resident).  In LDS (64 x 8 N floats = 32 KB per
  // wavefront at 16 cameras) four wavefronts fit a CU and the launch took 247 us for the ~8 k listed joints of a 98 304-frame 16cam6p batch;
  // from global memory with 16 wavefronts per CU 136 us (profiles/r4/README.md).
  const X32Mat A{wsw + lane, 64, 2 * N};
  __syncthreads();
  for (uint32_t it = blockIdx.x; it < cnt; it += gridDim.x) {
    const uint32_t raw_en = D.tri_list[it];
    if (only_marked && !(raw_en & 0x40000000u)) continue;   // handled by se3d_k_triangulate_exact4 (uniform: one entry per wavefront)
    const uint32_t en = raw_en & 0x3fffffffu;   // (bit 31: set below when the covariance comes from the fp64 sample path; bit 30: the four-joint kernel's hand-over mark)
    const int e = (int)(en >> 5), j = (int)(en & 31u), f = e / H, h = e - f * H;
    se3d_keypoint3d *kpo = &D.hypP[(size_t)f * H + h].keypoints[c_kp2fusion[D.d.pose_method][j]];
    // ---- usable views of the joint (conf >= 0.30, S3D:724-732), ascending camera order; row N on the fly with the reference's expressions ----
    const int32_t *arow = D.assoc + ((size_t)f * H + h) * N;
    int n = 0;
    for (int c0 = 0; c0 < N; c0 += 64) {
      const int c = c0 + lane;
      bool ok = false;
      se3d_keypoint2d r;
      r.x = r.y = 0.f; r.score = -1.f; r.cov[0] = r.cov[1] = r.cov[2] = 0.f;
      if (c < N) {
        const int k = arow[c];
        if (k >= 0) {
          const size_t cb = ((size_t)(f * N + c)) * D.d.P;
          r = D.raw[cb + D.orig[cb + k]].keypoints[j];
          ok = r.score >= TRI_THRESH;
        }
      }
      const unsigned long long bl = __ballot(ok);
      if (ok) {
        const int pos = n + __popcll(bl & ((1ull << lane) - 1ull));
        const float fx = D.K[c * 4 + 0], fy = D.K[c * 4 + 1], cx = D.K[c * 4 + 2], cy = D.K[c * 4 + 3];
        T.x[pos] = (r.x - cx) / fx; T.y[pos] = (r.y - cy) / fy; T.cf[pos] = r.score;
        T.k00[pos] = r.cov[0] / (fx * fx); T.k01[pos] = r.cov[1] / (fx * fy); T.k11[pos] = r.cov[2] / (fy * fy);   // S3D:324-326
        T.cam[pos] = c;
      }
      n += __popcll(bl);
    }
    __syncthreads();
    if (n < 2) {   // cannot happen for a listed joint; keep the slot defined
      se3d_zero_kp_words(kpo, lane, 64);
      continue;
    }
    for (int v = lane; v < n; v += 64) {
      float r0[4], r1[4];
      x32_dlt_rows(sRt + 12 * T.cam[v], T.x[v], T.y[v], r0, r1);
#pragma unroll
      for (int k = 0; k < 4; ++k) { T.rows[8 * v + k] = r0[k]; T.rows[8 * v + 4 + k] = r1[k]; }
    }
    __syncthreads();
    // ---- phase A: the weighted system (lane 0) and, with three or more views, its leave-one-view-out subsets (lanes 1 .. n) ----
    // ... and on lane n + 1 the unweighted system of all views: the centre sample of the unscented transform if no view is removed
    float Xl[3] = {0.f, 0.f, 0.f};
    double el = 1e300;
    // (joints seen by more than 62 views: no speculation — their candidates would not fit one pass anyway and are solved below when the
    // error asks for them; the centre sample takes lane 1)
    const bool spec = n >= 3 && n <= 62;
    const int centre_at = spec ? n + 1 : 1;
    if (lane == 0 || (spec && lane <= n) || lane == centre_at) {
      const bool wj = lane != centre_at;
      x32_triangulate(T, sRt, n, wj, wj ? lane - 1 : -1, -1, 0.f, 0.f, A, Xl);
      if (wj) el = x32_reproj_error(T, sRt, n, lane - 1, Xl);
    }
    float X = __shfl(Xl[0], 0), Y = __shfl(Xl[1], 0), Z = __shfl(Xl[2], 0);
    double err = se3d_readlane_f64(el, 0);
    float avg_score = 0.f;
    for (int v = 0; v < n; ++v) avg_score += T.cf[v];   // S3D:730
    avg_score /= (float)n;                               // S3D:738
    int rm = -1;
    if (err > REPROJ_MAX && n == 3) {  // S3D:748-792: the pair of remaining views with the smallest squared symmetric epipolar distance
      float dcand = 0.f;
      if (lane < 3) {
        const int i0 = (lane == 0) ? 1 : 0, i1 = (lane == 2) ? 1 : 2;
        const float p1x = T.x[i0], p1y = T.y[i0], p2x = T.x[i1], p2y = T.y[i1];
        const float *F = D.F + 9 * (size_t)se3d_pair_index(T.cam[i0], T.cam[i1], N);
        const float l1x = F[0] * p1x + (F[1] * p1y + F[2]);
        const float l1y = F[3] * p1x + (F[4] * p1y + F[5]);
        const float l1z = F[6] * p1x + (F[7] * p1y + F[8]);
        const float l2x = F[0] * p2x + (F[3] * p2y + F[6]);
        const float l2y = F[1] * p2x + (F[4] * p2y + F[7]);
        const float l2z = F[2] * p2x + (F[5] * p2y + F[8]);
        const float n1 = p2x * l1x + (p2y * l1y + l1z);
        const float n2 = p1x * l2x + (p1y * l2y + l2z);
        dcand = n1 * n1 / (l1x * l1x + l1y * l1y) + n2 * n2 / (l2x * l2x + l2y * l2y);
      }
      float best = (float)(err * err);
      for (int i = 0; i < 3; ++i) {
        const float di = __shfl(dcand, i);
        if (di < best) { best = di; rm = i; }
      }
      if (rm != -1) {   // re-triangulation of the two remaining views = the leave-one-out solve of lane rm + 1
        X = __shfl(Xl[0], rm + 1); Y = __shfl(Xl[1], rm + 1); Z = __shfl(Xl[2], rm + 1);
        err = se3d_readlane_f64(el, rm + 1);
        const int a = rm == 0 ? 1 : 0, b = rm == 2 ? 1 : 2;
        avg_score = (T.cf[a] + T.cf[b]) / 2.0f;
      }
    } else if (err > REPROJ_MAX && n >= 4) {  // S3D:793-838: candidates in ascending view order, first strict minimum below 0.9 x err wins
      const double err0 = err;
      double bestErr = err;
      for (int base = spec ? -1 : 0; base < n; base += 64) {
        // candidate base + lane: with speculation the one pass has the full system on lane 0 and candidates 0 .. n - 1 on lanes 1 .. n (solved
        // above); without it (n > 62 views) the passes solve their candidates now
        const int i = base + lane;
        double e_i = 1e300;
        float cx_ = 0.f, cy_ = 0.f, cz_ = 0.f;
        if (base == -1) {
          if (lane >= 1 && lane <= n) e_i = el;
          cx_ = Xl[0]; cy_ = Xl[1]; cz_ = Xl[2];
        } else if (i < n) {
          float Xc[3];
          x32_triangulate(T, sRt, n, true, i, -1, 0.f, 0.f, A, Xc);
          e_i = x32_reproj_error(T, sRt, n, i, Xc);
          cx_ = Xc[0]; cy_ = Xc[1]; cz_ = Xc[2];
        }
        if (!(e_i < 0.9 * err0)) e_i = 1e300;
        double m = e_i;
        int mi = i;
        for (int o = 32; o > 0; o >>= 1) {
          const double om = __shfl_xor(m, o);
          const int oi = __shfl_xor(mi, o);
          if (om < m || (om == m && oi < mi)) { m = om; mi = oi; }
        }
        if (m < bestErr) {
          bestErr = m;
          rm = mi;
          X = __shfl(cx_, mi - base); Y = __shfl(cy_, mi - base); Z = __shfl(cz_, mi - base);
        }
      }
      if (rm != -1) {
        err = bestErr;
        float tmpScore = 0.f;
        for (int v = 0; v < n; ++v)
          if (v != rm) tmpScore += T.cf[v];
        avg_score = tmpScore / (float)(n - 1);
      }
    }
    __syncthreads();
    if (rm != -1) {   // erase the view, order preserved
      if (lane == 0) {
        for (int v = rm; v + 1 < n; ++v) {
          T.x[v] = T.x[v + 1]; T.y[v] = T.y[v + 1]; T.cf[v] = T.cf[v + 1];
          T.k00[v] = T.k00[v + 1]; T.k01[v] = T.k01[v + 1]; T.k11[v] = T.k11[v + 1]; T.cam[v] = T.cam[v + 1];
          for (int k = 0; k < 8; ++k) T.rows[8 * v + k] = T.rows[8 * (v + 1) + k];
        }
      }
      n -= 1;
      __syncthreads();
    }
    if (err > REPROJ_MAX) avg_score = (float)((double)avg_score * (REPROJ_MAX / err));  // S3D:840-841
    // ---- phase B: unscented transform (S3D:471-523) ----
    // Round 6: WHICH arithmetic the 4n' + 1 sample solves need is decided here, on the kept views.  A joint is listed because its published point
    // or a decision of the outlier rejection needs the float32 sequence (phase A above: done) — its covariance needs float32 SAMPLES only when the
    // kept views' systems are ill conditioned (se3d_kappa_flag, the first pass's own test) or the covariance comes out wide / a sample does not
    // settle (SE3D_EXACT_VAR_MAX: those samples' systems are ill conditioned whatever the base system is).  Everything else — joints listed for the
    // reprojection-error band, for an outlier that phase A has just removed, for a main solve that took too many steps — gets the fast path's
    // fp64 sample solves around the float32 point (S3D:847: calc_covariance is handed the weighted result), i.e. the covariance contract of every
    // joint that never left the fast path (tests/parity.py cov_tolerance), at ~1/30 of the instructions.  Such an entry gets bit 31 set in tri_list.
    const int dim = 2 * n, nsamp = 4 * n + 1;
    const float bsc = sqrtf((float)dim + 0.5f);
    bool not_spd = false;
    for (int v = lane; v < n; v += 64) {
      const float l11 = sqrtf(T.k00[v]), l21 = T.k01[v] / l11, l22 = sqrtf(T.k11[v] - l21 * l21);
      not_spd |= !(isfinite(l11) && isfinite(l21) && isfinite(l22));
    }
    not_spd = __ballot(not_spd) != 0ull;
    float cov6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool cov64 = false;
    if (!not_spd && !cov32) {
      double gu_[10], gw_[10];
#pragma unroll
      for (int k = 0; k < 10; ++k) { gu_[k] = 0.0; gw_[k] = 0.0; }
      for (int v = lane; v < n; v += 64) {
        float r0[4], r1[4], w0[4], w1[4];
        const float cf = T.cf[v];
#pragma unroll
        for (int k = 0; k < 4; ++k) { r0[k] = T.rows[8 * v + k]; r1[k] = T.rows[8 * v + 4 + k]; w0[k] = r0[k] * cf; w1[k] = r1[k] * cf; }
        se3d_gram_add(gu_, r0); se3d_gram_add(gu_, r1);
        se3d_gram_add(gw_, w0); se3d_gram_add(gw_, w1);
      }
      float gwf[10], guf[10];
#pragma unroll
      for (int k = 0; k < 10; ++k) { gu_[k] = se3d_wave_sum(gu_[k]); gw_[k] = se3d_wave_sum(gw_[k]); guf[k] = (float)gu_[k]; gwf[k] = (float)gw_[k]; }
      if (!(se3d_kappa_flag(gwf, X, Y, Z) || se3d_kappa_flag(guf, X, Y, Z))) {   // uniform: every lane holds the same sums and the same point
        const float wgt = 1.0f / (2.0f * ((float)dim + 0.5f));
        double cv[6] = {0., 0., 0., 0., 0., 0.};
        bool sample_bad = false;
        const double vstart[4] = {(double)X, (double)Y, (double)Z, 1.0};
        // samples 0 .. 2n: centre + class A (both rows rebuilt); 2n + 1 .. 4n: class B (r1 alone) — se3d_sigma_gram, as in se3d_tri_joint_body
        for (int s0 = 0; s0 < nsamp; s0 += 64) {
          const int s = s0 + lane;
          if (s < nsamp) {
            double g[10];
#pragma unroll
            for (int k = 0; k < 10; ++k) g[k] = gu_[k];
            if (s > 0) {
              const bool clsB = s > 2 * n;
              const int q = clsB ? s - 2 * n - 1 : s - 1, c = q >> 1;
              float o0[4], o1[4], l11, l21, l22;
#pragma unroll
              for (int k = 0; k < 4; ++k) { o0[k] = T.rows[8 * c + k]; o1[k] = T.rows[8 * c + 4 + k]; }
              se3d_chol2(T.k00[c], T.k01[c], T.k11[c], l11, l21, l22);
              se3d_sigma_gram(g, sRt + 12 * T.cam[c], T.x[c], T.y[c], l11, l21, l22, bsc, o0, o1, (clsB ? 3 : 1) + (q & 1));
            }
            double pt[3];
            sample_bad |= !se3d_null_invit_point(g, vstart, pt);
            const float sx = (float)pt[0] - X, sy = (float)pt[1] - Y, sz = (float)pt[2] - Z;
            const double wx = (double)(sx * wgt), wy = (double)(sy * wgt), wz = (double)(sz * wgt);
            cv[0] += wx * sx; cv[1] += wx * sy; cv[2] += wx * sz; cv[3] += wy * sy; cv[4] += wy * sz; cv[5] += wz * sz;
          }
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) cv[k] = se3d_wave_sum(cv[k]);
        if (__ballot(sample_bad) == 0ull && cv[0] < SE3D_EXACT_VAR_MAX && cv[3] < SE3D_EXACT_VAR_MAX && cv[5] < SE3D_EXACT_VAR_MAX) {   // uniform
          cov64 = true;
#pragma unroll
          for (int k = 0; k < 6; ++k) cov6[k] = (float)cv[k];
        }
      }
    }
    if (lane == 0) D.tri_list[it] = cov64 ? (en | 0x80000000u) : en;   // (clears the hand-over mark)
    if (cov64) {
    } else if (not_spd) {   // defined policy: all entries NaN, position and score unaffected
#pragma unroll
      for (int k = 0; k < 6; ++k) cov6[k] = __builtin_nanf("");
    } else {
      // float32 samples, one per lane (S3D:471-518); the weighted outer products are summed in sample order in float (S3D:521-522)
      // sample 0 (the unmoved points) was solved in phase A (lane centre_at); it is valid when no view was removed
      const int first = (rm == -1) ? 1 : 0;
      if (first) {
        const float c0 = __shfl(Xl[0], centre_at), c1 = __shfl(Xl[1], centre_at), c2 = __shfl(Xl[2], centre_at);
        if (lane == 0) { cen[0] = c0 - X; cen[1] = c1 - Y; cen[2] = c2 - Z; }
      }
      for (int s0 = first; s0 < nsamp; s0 += 64) {
        const int s = s0 + lane;
        if (s < nsamp) {
          int mv = -1;
          float mx = 0.f, my = 0.f;
          if (s > 0) {   // mod_samples (S3D:471-487): samples 4c + 1 .. 4c + 4 move view c
            const int c = (s - 1) >> 2, kind = (s - 1) & 3;
            const float l11 = sqrtf(T.k00[c]), l21 = T.k01[c] / l11, l22 = sqrtf(T.k11[c] - l21 * l21);
            const float dx1 = l11 * bsc, dy1 = l21 * bsc, dy2 = l22 * bsc;
            mv = c; mx = T.x[c]; my = T.y[c];
            if (kind == 0) { mx = mx - dx1; my = my - dy1; }
            else if (kind == 1) { my = my - dy2; }
            else if (kind == 2) { mx = mx + dx1; my = my + dy1; }
            else { my = my + dy2; }
          }
          float Xs[3];
          x32_triangulate(T, sRt, n, false, -1, mv, mx, my, A, Xs);
          cen[3 * s + 0] = Xs[0] - X; cen[3 * s + 1] = Xs[1] - Y; cen[3 * s + 2] = Xs[2] - Z;
        }
      }
      __syncthreads();
      const float wgt = 1.0f / (2.0f * ((float)dim + 0.5f));   // weights[0] = 2 kappa = 1, all / (2 (dim + kappa))
      if (lane < 6) {   // (centred * weights) * centred^T, summed over the samples in order (S3D:521-522)
        const int ci = lane < 3 ? 0 : (lane < 5 ? 1 : 2), cj = lane < 3 ? lane : (lane < 5 ? lane - 2 : 2);
        float acc = 0.f;
        for (int s = 0; s < nsamp; ++s) acc += (cen[3 * s + ci] * wgt) * cen[3 * s + cj];
        cov6[0] = acc;
      }
#pragma unroll
      for (int k = 5; k >= 0; --k) cov6[k] = __shfl(cov6[0], k);
    }
    if (lane == 0) {
      kpo->joint[0] = (double)X; kpo->joint[1] = (double)Y; kpo->joint[2] = (double)Z;
      kpo->score = avg_score; kpo->_pad = 0.f;
#pragma unroll
      for (int k = 0; k < 6; ++k) kpo->cov[k] = (double)cov6[k];
    }
    __syncthreads();
  }
}

// =============================================================================================================
// rows X, D — per-person plausibility post-processing (S3D:861-973), duplicate merge (S3D:984-996) and output
// compaction.  One workgroup per frame.
// =============================================================================================================
__device__ __forceinline__ double se3d_jdist(const double a[3], const double b[3]) {
  return sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]));
}
__device__ __forceinline__ void se3d_addcov(se3d_keypoint3d &kp, double sigma) {
  kp.cov[0] += sigma * sigma; kp.cov[3] += sigma * sigma; kp.cov[5] += sigma * sigma;
}
__device__ __forceinline__ void se3d_zero_kp(se3d_keypoint3d &kp) {
  kp.joint[0] = kp.joint[1] = kp.joint[2] = 0.0; kp.score = 0.f; kp._pad = 0.f;
  for (int i = 0; i < 6; ++i) kp.cov[i] = 0.0;
}

// per-person plausibility post-processing (S3D:861-973) on one person (global or LDS); returns the surviving keypoint count
__device__ __forceinline__ int se3d_post_filter(se3d_person3d &p, int pm) {
    int num_valid = 0;
    for (int k = 0; k < SE3D_NKP; ++k) num_valid += (p.keypoints[c_kp2fusion[pm][k]].score > 0.f) ? 1 : 0;
    for (int k = 0; k < SE3D_NKP; ++k) {  // limb-length covariance inflation (S3D:861-883)
      se3d_keypoint3d &kp = p.keypoints[c_kp2fusion[pm][k]];
      if (kp.score <= 0.f) continue;
      const int parent = c_kpParent[pm][k];
      if (parent >= 0) {
        const se3d_keypoint3d &pk = p.keypoints[c_kp2fusion[pm][parent]];
        if (pk.score > 0.f && c_limbLength[pm][k] > 0) {
          const double jd = se3d_jdist(kp.joint, pk.joint);
          se3d_addcov(kp, 0.075 * (jd - c_limbLength[pm][k]) / c_limbLSigma[pm][k]);
        } else if (pm == SE3D_POSE_SIMPLE && k == 6) {  // RShoulder special case (S3D:875-881)
          se3d_keypoint3d &ls = p.keypoints[c_kp2fusion[pm][5]];
          if (ls.score > 0.f) {
            const double jd = se3d_jdist(kp.joint, ls.joint);
            se3d_addcov(kp, 0.075 * (jd - 0.35) / 0.15);
            se3d_addcov(ls, 0.075 * (jd - 0.35) / 0.15);
          }
        }
      }
    }
    double root[3] = {0, 0, 0};
    float root_score = 0.f;
    if (p.keypoints[FU_MidHip].score > 0.f) {
      for (int a = 0; a < 3; ++a) root[a] = p.keypoints[FU_MidHip].joint[a];
      root_score = p.keypoints[FU_MidHip].score;
    } else if (p.keypoints[FU_LHip].score > 0.f && p.keypoints[FU_RHip].score > 0.f) {
      for (int a = 0; a < 3; ++a) root[a] = (p.keypoints[FU_LHip].joint[a] + p.keypoints[FU_RHip].joint[a]) / 2.;
      root_score = (p.keypoints[FU_LHip].score + p.keypoints[FU_RHip].score) / 2.f;
    }
    if (root_score > 0.f) {  // S3D:937-953
      for (int k = 0; k < SE3D_NFU; ++k) {
        se3d_keypoint3d &kp = p.keypoints[k];
        if (kp.score > 0.f) {
          if (se3d_jdist(root, kp.joint) > 2.0) { se3d_zero_kp(kp); --num_valid; }
        } else {
          se3d_zero_kp(kp);
          --num_valid;
        }
      }
    }
    double feet = 0.0;  // S3D:955-966
    if (p.keypoints[FU_LAnkle].score > 0.f && p.keypoints[FU_RAnkle].score > 0.f)
      feet = (p.keypoints[FU_LAnkle].joint[2] + p.keypoints[FU_RAnkle].joint[2]) / 2.0;
    else if (p.keypoints[FU_LAnkle].score > 0.f) feet = p.keypoints[FU_LAnkle].joint[2];
    else if (p.keypoints[FU_RAnkle].score > 0.f) feet = p.keypoints[FU_RAnkle].joint[2];
    if (fabs(feet) > 0.50) num_valid = 0;
    return num_valid;
}

#define SE3D_POST_SMALL_H 16
// Small rigs (H <= 16): one wavefront per frame.  The plausibility filters (S3D:861-966) and the duplicate-merge distances (S3D:392-408) run over
// (person, joint) lanes; whatever is ORDER-sensitive stays in the reference's order — a keypoint's own limb-length term before the right-shoulder
// special case adds to the left shoulder (S3D:875-881), the root copied before any keypoint is removed (S3D:937-953), the mean joint distance
// of two persons summed joint by joint by one lane (only the square roots under the sum are parallel).  The persons stay in GLOBAL memory
// (hypP, modified in place: it is an intermediate between triangulation and this kernel) and LDS holds only what lanes read from EACH OTHER —
// joint position and score per (person, joint): 32 B instead of the 80-B keypoint, 6.7 KB per wavefront at H = 8, 23 wavefronts per CU on a
// kernel that sits in s_waitcnt three quarters of the time.  A removed keypoint is zeroed in hypP and in LDS, a merge rewrites person i in
// both; the published persons are then a hypP -> outP copy behind a fence.
// The limb-length covariance inflation — which touches three words of nearly every keypoint — is not written back into hypP: its terms wait
// in LDS (one double per (person, joint) plus the left shoulder's second term per person) and are added, in the reference's order, where a
// covariance is consumed: by a merge, or by the copy into outP.  hypP is then written only by the rare removals and merges instead of having
// every line dirtied (~1 GB of write-back per 98 304-frame launch).  History (per 98 304 frames at 16cam6p): one lane per person 2.24 ms, (person,
// joint) lanes with the persons staged in LDS 1.51 ms, persons in place 1.07 ms, inflation deferred 1.02 ms (profiles/r2).  Published persons
// bytewise equal to the block kernel below (`post_form` option, tests/test_gpu_stage1.py).
struct PostKp { double joint[3]; float score; float _pad; };
__global__ __launch_bounds__(64) void se3d_k_postprocess_wave3(Se3dDev D) {
  extern __shared__ __align__(16) unsigned char post_smem[];
  const int f = blockIdx.x, lane = threadIdx.x;
  const int N = D.d.N, P = D.d.P, H = D.d.H;
  const int pm = D.d.pose_method;
  PostKp *sk = reinterpret_cast<PostKp *>(post_smem);                                 // [H][21]
  double *term = reinterpret_cast<double *>(sk + (size_t)H * SE3D_NFU);               // [H][21]
  int *cnt0 = reinterpret_cast<int *>(term + (size_t)H * SE3D_NFU), *dec = cnt0 + H;  // [H] valid input keypoints / removed fusion slots
  double *infl = reinterpret_cast<double *>(dec + H);                                 // [H][21] own limb term^2, then [H] the left shoulder's second term^2
  double *infl2 = infl + (size_t)H * SE3D_NFU;
  const int nh = D.nhyp[f];
  se3d_person3d *persons = D.hypP + (size_t)f * H;
  constexpr int words2 = (int)(sizeof(se3d_person3d) / 8);
  int ncam = 0;
  for (int c = lane; c < N; c += 64) ncam += (D.nraw[f * N + c] > 0) ? 1 : 0;
  for (int o = 32; o > 0; o >>= 1) ncam += __shfl_xor(ncam, o);
  const bool few = ncam < 2;  // S3D:557-560
  for (int it = lane; it < nh * SE3D_NFU; it += 64) {
    const int h = it / SE3D_NFU, k = it - h * SE3D_NFU;
    const se3d_keypoint3d &g = persons[h].keypoints[k];
    PostKp c;
    c.joint[0] = g.joint[0]; c.joint[1] = g.joint[1]; c.joint[2] = g.joint[2]; c.score = g.score; c._pad = 0.f;
    sk[it] = c;
    infl[it] = 0.0;
  }
  const int nv_l = (lane < nh) ? D.nviews[f * H + lane] : 0;
  if (lane < H) { cnt0[lane] = 0; dec[lane] = 0; infl2[lane] = 0.0; }
  __syncthreads();
  // ---- limb-length covariance inflation (S3D:861-883): items (person, input keypoint); the owner lane adds every term of its keypoint ----
  for (int it = lane; it < nh * SE3D_NKP; it += 64) {
    const int h = it / SE3D_NKP, k = it - h * SE3D_NKP;
    const PostKp *pk_ = sk + (size_t)h * SE3D_NFU;
    const int fu = c_kp2fusion[pm][k];
    const PostKp &kp = pk_[fu];
    if (kp.score <= 0.f) continue;
    atomicAdd(&cnt0[h], 1);
    double s_own = 0.0, s_56 = 0.0;   // sigma of the keypoint's own limb term / of the right shoulder's term for the LEFT shoulder
    bool has_own = false, has_56 = false;
    const int parent = c_kpParent[pm][k];
    if (parent >= 0) {
      const PostKp &pk = pk_[c_kp2fusion[pm][parent]];
      if (pk.score > 0.f && c_limbLength[pm][k] > 0) {
        s_own = 0.075 * (se3d_jdist(kp.joint, pk.joint) - c_limbLength[pm][k]) / c_limbLSigma[pm][k];
        has_own = true;
      } else if (pm == SE3D_POSE_SIMPLE && k == 6) {  // RShoulder special case (S3D:875-881): its own share
        const PostKp &ls = pk_[c_kp2fusion[pm][5]];
        if (ls.score > 0.f) { s_own = 0.075 * (se3d_jdist(kp.joint, ls.joint) - 0.35) / 0.15; has_own = true; }
      }
    }
    if (pm == SE3D_POSE_SIMPLE && k == 5) {  // ... and the left shoulder's share of it, added after its own term (k = 5 precedes k = 6)
      const PostKp &rs = pk_[c_kp2fusion[pm][6]];
      const int par6 = c_kpParent[pm][6];
      const bool rs_case1 = par6 >= 0 && pk_[c_kp2fusion[pm][par6]].score > 0.f && c_limbLength[pm][6] > 0;
      if (rs.score > 0.f && par6 >= 0 && !rs_case1) { s_56 = 0.075 * (se3d_jdist(rs.joint, kp.joint) - 0.35) / 0.15; has_56 = true; }
    }
    if (has_own) infl[(size_t)h * SE3D_NFU + fu] = s_own * s_own;
    if (has_56) infl2[h] = s_56 * s_56;     // only the left shoulder (fusion slot 5) has a second term
  }
  // ---- distance to the root (S3D:937-953): items (person, fusion slot); every lane reads the root before anything is removed ----
  for (int it = lane; it < nh * SE3D_NFU; it += 64) {
    const int h = it / SE3D_NFU, k = it - h * SE3D_NFU;
    const PostKp *p = sk + (size_t)h * SE3D_NFU;
    double root[3] = {0, 0, 0};
    float root_score = 0.f;
    if (p[FU_MidHip].score > 0.f) {
      for (int a = 0; a < 3; ++a) root[a] = p[FU_MidHip].joint[a];
      root_score = p[FU_MidHip].score;
    } else if (p[FU_LHip].score > 0.f && p[FU_RHip].score > 0.f) {
      for (int a = 0; a < 3; ++a) root[a] = (p[FU_LHip].joint[a] + p[FU_RHip].joint[a]) / 2.;
      root_score = (p[FU_LHip].score + p[FU_RHip].score) / 2.f;
    }
    bool rm = false;
    if (root_score > 0.f) rm = !(p[k].score > 0.f) || se3d_jdist(root, p[k].joint) > 2.0;
    term[it] = rm ? 1.0 : 0.0;
  }
  __syncthreads();
  for (int it = lane; it < nh * SE3D_NFU; it += 64)
    if (term[it] != 0.0) {
      const int h = it / SE3D_NFU, k = it - h * SE3D_NFU;
      se3d_zero_kp(persons[h].keypoints[k]);
      PostKp z; z.joint[0] = z.joint[1] = z.joint[2] = 0.0; z.score = 0.f; z._pad = 0.f;
      sk[it] = z;
      { infl[it] = 0.0; if (k == FU_LShoulder) infl2[h] = 0.0; }   // the zeroed keypoint carries no inflation
      atomicAdd(&dec[h], 1);
    }
  __syncthreads();
  bool al = false;
  if (lane < nh && !few && nv_l >= 2) {
    const PostKp *p = sk + (size_t)lane * SE3D_NFU;
    int num_valid = cnt0[lane] - dec[lane];
    double feet = 0.0;  // S3D:955-966
    if (p[FU_LAnkle].score > 0.f && p[FU_RAnkle].score > 0.f) feet = (p[FU_LAnkle].joint[2] + p[FU_RAnkle].joint[2]) / 2.0;
    else if (p[FU_LAnkle].score > 0.f) feet = p[FU_LAnkle].joint[2];
    else if (p[FU_RAnkle].score > 0.f) feet = p[FU_RAnkle].joint[2];
    if (fabs(feet) > 0.50) num_valid = 0;
    al = num_valid > 9;
  }
  unsigned long long alive = __ballot(al);
  // ---- duplicate merge (S3D:984-996): distances from the LDS copy ----
  for (int i = 0; i < nh; ++i) {
    if (!((alive >> i) & 1ull)) continue;
    int from = i + 1;
    for (;;) {
      if (!(alive >> from)) break;
      const int nj = nh - from;
      for (int it = lane; it < nj * SE3D_NFU; it += 64) {
        const int jo = it / SE3D_NFU, k = it - jo * SE3D_NFU, jj = from + jo;
        double t = -1.0;
        if ((alive >> jj) & 1ull) {
          const PostKp &a = sk[(size_t)i * SE3D_NFU + k], &bb = sk[(size_t)jj * SE3D_NFU + k];
          if (a.score > 0.f && bb.score > 0.f) {
            const double dx = a.joint[0] - bb.joint[0], dy = a.joint[1] - bb.joint[1], dz = a.joint[2] - bb.joint[2];
            t = sqrt(dx * dx + dy * dy + dz * dz);
          }
        }
        term[it] = t;
      }
      __syncthreads();
      bool hit = false;
      if (lane < nj && ((alive >> (from + lane)) & 1ull)) {
        int nu = 0;
        double dist = 0.0;
        for (int k = 0; k < SE3D_NFU; ++k) {
          const double t = term[lane * SE3D_NFU + k];
          if (t >= 0.0) { dist += t; ++nu; }
        }
        hit = (nu > 0 ? dist / nu : MAX_COSTS) < 0.20;
      }
      const unsigned long long hb = __ballot(hit);
      if (!hb) break;
      const int jm = from + __ffsll((long long)hb) - 1;
      __syncthreads();
      if (lane < SE3D_NFU) {  // merge_persons (S3D:410-423), one joint per lane: hypP (all fields) and the LDS copy (position, score)
        se3d_keypoint3d &k1 = persons[i].keypoints[lane];
        const se3d_keypoint3d &k2 = persons[jm].keypoints[lane];
        PostKp &c1 = sk[(size_t)i * SE3D_NFU + lane];
        const PostKp &c2 = sk[(size_t)jm * SE3D_NFU + lane];
        const double total = (double)(c1.score + c2.score);
        if (total > 0.0) {
          for (int a = 0; a < 3; ++a) { c1.joint[a] = ((double)c1.score * c1.joint[a] + (double)c2.score * c2.joint[a]) / total; k1.joint[a] = c1.joint[a]; }
          c1.score = fmaxf(c1.score, c2.score);
          k1.score = c1.score;
          double v1[6], v2[6];
          for (int a = 0; a < 6; ++a) { v1[a] = k1.cov[a]; v2[a] = k2.cov[a]; }
          {   // the pending inflation of both keypoints, in the reference's order (own term, then the left shoulder's second one)
            const double o1 = infl[(size_t)i * SE3D_NFU + lane], o2 = infl[(size_t)jm * SE3D_NFU + lane];
            const double s1 = lane == FU_LShoulder ? infl2[i] : 0.0, s2 = lane == FU_LShoulder ? infl2[jm] : 0.0;
            v1[0] = (v1[0] + o1) + s1; v1[3] = (v1[3] + o1) + s1; v1[5] = (v1[5] + o1) + s1;
            v2[0] = (v2[0] + o2) + s2; v2[3] = (v2[3] + o2) + s2; v2[5] = (v2[5] + o2) + s2;
            infl[(size_t)i * SE3D_NFU + lane] = 0.0;
            if (lane == FU_LShoulder) infl2[i] = 0.0;
          }
          for (int a = 0; a < 6; ++a) k1.cov[a] = (v1[a] + v2[a]) / 2.0;
        }
      }
      __threadfence_block();   // the merged covariances were formed from values other lanes wrote to hypP (inflation, removals) ...
      __syncthreads();
      alive &= ~(1ull << jm);
      from = jm + 1;
      if (from >= nh) break;
    }
  }
  // ---- compaction in ascending hypothesis order + association table in original detection indices ----
  __threadfence_block();   // ... and so is the copy below: hypP holds the modifications of every lane of this wavefront
  __syncthreads();
  const int total = __popcll(alive);
  int pos = 0;
  for (unsigned long long m = alive; m; m &= m - 1ull, ++pos) {
    const int h = __ffsll((long long)m) - 1;
    const uint2 *src = reinterpret_cast<const uint2 *>(&persons[h]);
    uint2 *dst = reinterpret_cast<uint2 *>(&D.outP[(size_t)f * H + pos]);
    for (int w = lane; w < words2; w += 64) {
      uint2 v = src[w];
      if (w >= 1) {   // word 0: id, score; keypoint kk: words 1 + 10 kk .. 10 + 10 kk = joint[3], (score, pad), cov[6]
        const int kk = (w - 1) / 10, r = (w - 1) - 10 * kk;
        if (r == 4 || r == 7 || r == 9) {   // cov[0], cov[3], cov[5]
          double c = __builtin_bit_cast(double, v);
          c = (c + infl[(size_t)h * SE3D_NFU + kk]) + (kk == FU_LShoulder ? infl2[h] : 0.0);
          v = __builtin_bit_cast(uint2, c);
        }
      }
      dst[w] = v;
    }
    if (lane == 0) D.out_hyp[(size_t)f * H + pos] = h;
  }
  if (lane == 0) {
    D.out_count[f] = total;
    if (few) {  // the reference returns before building any hypothesis (S3D:557-560)
      atomicOr(&D.flags[f], SE3D_FLAG_FEW_CAMERAS);
      D.nhyp[f] = 0;
    }
  }
  for (int i = lane; i < nh * N; i += 64) {
    const int h = i / N, c = i - h * N;
    const int k = D.assoc[((size_t)f * H + h) * N + c];
    D.out_assoc[((size_t)f * H + h) * N + c] = (k >= 0) ? D.orig[((size_t)(f * N + c)) * P + k] : -1;
  }
}
static size_t se3d_post_wave3_lds_bytes(int H) { return (size_t)H * SE3D_NFU * (sizeof(PostKp) + 8) + (size_t)H * 8 + (size_t)H * SE3D_NFU * 8 + (size_t)H * 8; }

__global__ __launch_bounds__(256) void se3d_k_postprocess(Se3dDev D) {
  const int f = blockIdx.x, tid = threadIdx.x;
  const int N = D.d.N, P = D.d.P, H = D.d.H;
  const int pm = D.d.pose_method;
  const int nh = D.nhyp[f];
  se3d_person3d *persons = D.hypP + (size_t)f * H;
  __shared__ int alive[SE3D_MUNKRES_MAX];
  __shared__ int s_ncam;
  __shared__ int wsum[4];
  if (tid == 0) s_ncam = 0;
  __syncthreads();
  {
    int local = 0;
    for (int c = tid; c < N; c += 256) local += (D.nraw[f * N + c] > 0) ? 1 : 0;
    if (local) atomicAdd(&s_ncam, local);
  }
  for (int i = tid; i < SE3D_MUNKRES_MAX; i += 256) alive[i] = 0;
  __syncthreads();
  const bool few = s_ncam < 2;  // S3D:557-560
  for (int h = tid; h < nh && !few; h += 256) {
    if (D.nviews[f * H + h] < 2) continue;
    const int num_valid = se3d_post_filter(persons[h], pm);
    alive[h] = (num_valid > 9) ? 1 : 0;
  }
  __syncthreads();
  // ---- duplicate merge (S3D:984-996): sequential in i, distances to all later j in parallel ----
  __shared__ int s_first;
  for (int i = 0; i < nh; ++i) {
    if (!alive[i]) continue;  // uniform (LDS)
    int from = i + 1;
    for (;;) {
      if (tid == 0) s_first = 0x7fffffff;
      __syncthreads();
      const int jj = from + tid;  // nh <= 256 = blockDim
      if (jj < nh && alive[jj]) {
        int nu = 0;
        double dist = 0.0;
        for (int k = 0; k < SE3D_NFU; ++k) {
          const se3d_keypoint3d &a = persons[i].keypoints[k], &bb = persons[jj].keypoints[k];
          if (a.score > 0.f && bb.score > 0.f) {
            const double dx = a.joint[0] - bb.joint[0], dy = a.joint[1] - bb.joint[1], dz = a.joint[2] - bb.joint[2];
            dist += sqrt(dx * dx + dy * dy + dz * dz);
            ++nu;
          }
        }
        const double dd = nu > 0 ? dist / nu : MAX_COSTS;
        if (dd < 0.20) atomicMin(&s_first, jj);
      }
      __syncthreads();
      const int jm = s_first;
      __syncthreads();
      if (jm == 0x7fffffff) break;
      if (tid < SE3D_NFU) {  // merge_persons (S3D:410-423), one joint per thread
        se3d_keypoint3d &k1 = persons[i].keypoints[tid];
        const se3d_keypoint3d &k2 = persons[jm].keypoints[tid];
        const double total = (double)(k1.score + k2.score);
        if (total > 0.0) {
          for (int a = 0; a < 3; ++a) k1.joint[a] = ((double)k1.score * k1.joint[a] + (double)k2.score * k2.joint[a]) / total;
          k1.score = fmaxf(k1.score, k2.score);
          for (int a = 0; a < 6; ++a) k1.cov[a] = (k1.cov[a] + k2.cov[a]) / 2.0;
        }
      }
      if (tid == 0) alive[jm] = 0;
      from = jm + 1;
      __syncthreads();
    }
  }
  __syncthreads();
  // ---- compaction in ascending hypothesis order + association table in original detection indices ----
  int total;
  const bool al = (tid < nh) && alive[tid];
  const int pos = se3d_block_excl_scan(al, wsum, total);
  if (al) {
    D.out_hyp[(size_t)f * H + pos] = tid;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(&persons[tid]);
    uint32_t *dst = reinterpret_cast<uint32_t *>(&D.outP[(size_t)f * H + pos]);
    for (int w = 0; w < (int)(sizeof(se3d_person3d) / 4); ++w) dst[w] = src[w];
  }
  if (tid == 0) {
    D.out_count[f] = total;
    if (few) {  // the reference returns before building any hypothesis (S3D:557-560)
      atomicOr(&D.flags[f], SE3D_FLAG_FEW_CAMERAS);
      D.nhyp[f] = 0;
    }
  }
  for (int i = tid; i < nh * N; i += 256) {
    const int h = i / N, c = i - h * N;
    const int k = D.assoc[((size_t)f * H + h) * N + c];
    D.out_assoc[((size_t)f * H + h) * N + c] = (k >= 0) ? D.orig[((size_t)(f * N + c)) * P + k] : -1;
  }
}

// Single-wavefront sweep for H, P <= 64: same decisions as se3d_sweep_assign_body (seeding, feasibility counts, "last feasible
// detection wins", Hungarian on the unmasked costs only when ambiguous, attach / spawn order), expressed with ballots.
__device__ __forceinline__ void se3d_sweep_assign_wave(const Se3dDev &D, int b, int f, int lane, double *W) {
  const int N = D.d.N, P = D.d.P, H = D.d.H;
  if (D.nraw[f * N + b] == 0) return;  // S3D:538-555
  const int nh = D.nhyp[f];
  const int nd = D.nvalid[f * N + b];
  int32_t *assoc = D.assoc + (size_t)f * H * N;
  if (nh == 0) {  // seeding (S3D:565-586)
    if (lane < nd && lane < H) { assoc[(size_t)lane * N + b] = lane; D.nviews[f * H + lane] = 1; }
    if (lane == 0) D.nhyp[f] = min(nd, H);
    return;
  }
  if (nd == 0) return;
  const uint8_t *mask = D.mask + (size_t)f * H * P;
  int cnt = 0, assign = -1;
  if (lane < nh)
    for (int q = 0; q < nd; ++q)
      if (mask[lane + (size_t)H * q]) { ++cnt; assign = q; }   // S3D:620-623
  int ccnt = 0;
  if (lane < nd)
    for (int h = 0; h < nh; ++h) ccnt += mask[h + (size_t)H * lane];
  if (__ballot(cnt > 1 || ccnt > 1)) {  // S3D:628-634
    const double *C = D.C + (size_t)f * H * P;
    if (lane < nh)
      for (int q = 0; q < nd; ++q) W[lane + nh * q] = C[lane + (size_t)H * q];
    const int rc = se3d_munkres_wave(W, nh, nh, nd, lane, assign);
    if (rc != 0 && lane == 0) atomicOr(&D.flags[f], SE3D_FLAG_ASSIGNMENT_FAILED);
  }
  // attach or spawn (S3D:636-660)
  bool sp = false;
  const int det = (lane < nh) ? assign : -1;
  if (det >= 0) {
    if (mask[lane + (size_t)H * det]) {
      assoc[(size_t)lane * N + b] = det;
      D.nviews[f * H + lane] += 1;
    } else {
      sp = true;
    }
  }
  bool handled = false;
  for (int h = 0; h < nh; ++h) handled |= (__builtin_amdgcn_readlane(det, h) == lane);
  const unsigned long long lt = (1ull << lane) - 1ull;
  const unsigned long long spb = __ballot(sp);
  const int tot1 = __popcll(spb);
  if (sp) {
    const int nhh = nh + __popcll(spb & lt);
    if (nhh < H) { assoc[(size_t)nhh * N + b] = det; D.nviews[f * H + nhh] = 1; }
  }
  const bool un = (lane < nd) && !handled;  // S3D:662-673
  const unsigned long long unb = __ballot(un);
  const int tot2 = __popcll(unb);
  if (un) {
    const int nhh = nh + tot1 + __popcll(unb & lt);
    if (nhh < H) { assoc[(size_t)nhh * N + b] = lane; D.nviews[f * H + nhh] = 1; }
  }
  if (lane == 0) {
    const int nn = nh + tot1 + tot2;
    if (nn > H) atomicOr(&D.flags[f], SE3D_FLAG_HYP_OVERFLOW);
    D.nhyp[f] = min(nn, H);
  }
}
__global__ __launch_bounds__(64) void se3d_k_sweep_fused_wave(Se3dDev D) {
  extern __shared__ __align__(16) double s_dynW_w[];
  const int f = blockIdx.x, lane = threadIdx.x;
  for (int b = 0; b < D.d.N; ++b) {
    if (b > 0) {
      se3d_sweep_cost_small(D, b, f, lane, 64);
      __syncthreads();  // one wavefront: orders the global C / mask stores before the loads of other lanes
    }
    se3d_sweep_assign_wave(D, b, f, lane, s_dynW_w);
    __syncthreads();
  }
}

// stand-alone Munkres (se3d_hungarian): one workgroup, dist column-major with ld = nR
// The one-wavefront sweep with its working set in LDS (round 2).  se3d_k_sweep_fused_wave keeps the association table, the view counts,
// the cost matrix and the veto mask of the frame in global memory: every camera step is a chain of ~6 dependent L2 round trips (assoc ->
// pair table -> C / mask stores -> loads by other lanes -> attach / spawn stores -> next step), ~100 per frame, and the kernel sat at 78 %
// s_waitcnt with 17 % VALU issue utilisation.  Here only the pair-table reads (8 in flight per (hypothesis, detection) item) go to
// memory; assoc / nviews / nhyp are written back once at the end.  Same decisions in the same order as se3d_sweep_cost_small +
// se3d_sweep_assign_wave (S3D:538-673): the association tables of the three launch structures are compared bytewise in
// tests/test_gpu_stage1.py.
#ifdef SE3D_LM_STAMPS   // diagnostic build only (make diag, tools/ff_stamps.py): cycle stamps of the one-wavefront sweep
__device__ unsigned long long g_ff_stamps[16];
#define FF_STAMP_INIT unsigned long long ff_prev_ = __builtin_readcyclecounter();
#define FF_STAMP(idx) { const unsigned long long now_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) atomicAdd(&g_ff_stamps[idx], now_ - ff_prev_); ff_prev_ = now_; }
extern "C" void se3d_diag_ff_stamps(unsigned long long *out16, int reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_ff_stamps), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ff_stamps), z, sizeof(z)); }
}
#else
#define FF_STAMP_INIT
#define FF_STAMP(idx)
#endif
// LDS working set of the one-wavefront sweep (se3d_k_sweep_wave_lds)
struct SweepWaveLds {
  double *W, *C;        // [H * P] Munkres working matrix; cost, hypothesis-major like D.C
  float *psc;           // [N * P] detection scores of the frame by filtered slot
  int32_t *assoc;       // [H * N]
  int *nviews;          // [H]
  int *nraw, *nvalid;   // [N] each
  uint8_t *mask;        // [H * P]
  double *inv;          // [2 N + 1] RN(1 / k): the quotients by view counts (se3d_div_count and 1 / n_obs, 1 / (2 n_obs))
};
// The sweep of frame f by one wavefront; tab(a, k, b, q, nd): pair cost of (camera a, filtered slot k) x (camera b, filtered slot q), nd = valid
// detections of camera b.  L.psc / nraw / nvalid hold the frame's values, assoc = -1 and nviews = 0 on entry; every __syncthreads() below is
// reached by the one (surviving) wavefront of the workgroup only.
template <typename Tab>
__device__ __forceinline__ void se3d_sweep_wave_run(const Se3dDev &D, int f, int lane, const SweepWaveLds &L, Tab tab) {
  const int N = D.d.N, P = D.d.P, H = D.d.H;
  double *W = L.W, *C = L.C;
  float *psc = L.psc;
  int32_t *assoc = L.assoc;
  int *nviews = L.nviews, *nraw = L.nraw, *nvalid = L.nvalid;
  uint8_t *mask = L.mask;
  const double max_epi = D.d.max_epi;
  int nh = 0;   // hypotheses so far: every lane derives the same value from the ballots
  FF_STAMP_INIT
  for (int b = 0; b < N; ++b) {
    FF_STAMP(7)   // loop overhead / seeding
    if (nraw[b] == 0) continue;  // S3D:538-555
    const int nd = nvalid[b];
    if (nh == 0) {  // seeding (S3D:565-586)
      if (lane < nd && lane < H) { assoc[(size_t)lane * N + b] = lane; nviews[lane] = 1; }
      nh = min(nd, H);
      __syncthreads();
      continue;
    }
    if (nd == 0) continue;
    // ---- cost and veto mask of (hypothesis, detection of camera b): se3d_sweep_cost_small ----
    for (int idx = lane; idx < nh * nd; idx += 64) {
      const int h = idx % nh, q = idx / nh;
      const int32_t *arow = assoc + (size_t)h * N;
      const int n_obs = nviews[h];
      const double tolerance = 1.0 - L.inv[2 * n_obs], veto_delta = L.inv[n_obs];   // 1.0 / (2 * n_obs), 1.0 / n_obs (S3D:616-626): correctly rounded reciprocals
      double total = 0.0;
      int nused = 0, cnt = 0;
      for (int a0 = 0; a0 < b; a0 += 8) {  // 8 independent pair-table loads in flight, ordered accumulation
        int k[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) k[u] = (a0 + u < b) ? arow[a0 + u] : -1;
        double c[8];
        float sc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          c[u] = -1.0; sc[u] = 0.f;
          if (k[u] >= 0) {
            c[u] = tab(a0 + u, k[u], b, q, nd);
            sc[u] = psc[(size_t)(a0 + u) * P + k[u]];
          }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (c[u] >= 0.0) {
            total += c[u];
            ++nused;
            if (c[u] > max_epi && (sc[u] > 0.5f || n_obs == 1)) ++cnt;
          }
        }
      }
      bool veto = false;
      if (cnt > 0) {
        double tmp = 0.0;
        for (int i = 0; i < cnt; ++i) tmp += veto_delta;
        veto = tmp > tolerance;
      }
      double Cv;
      if (nused > 0) Cv = se3d_div_count(total, nused, L.inv[nused]);   // total / nused
      else { Cv = MAX_COSTS; veto = true; }
      C[h + (size_t)H * q] = Cv;
      mask[h + (size_t)H * q] = (!veto && Cv < max_epi) ? 1 : 0;
    }
    __syncthreads();
    FF_STAMP(4)   // cost + veto mask
    // ---- assignment, attach / spawn: se3d_sweep_assign_wave ----
    int cnt = 0, assign = -1;
    if (lane < nh)
      for (int q = 0; q < nd; ++q)
        if (mask[lane + (size_t)H * q]) { ++cnt; assign = q; }   // S3D:620-623
    int ccnt = 0;
    if (lane < nd)
      for (int h = 0; h < nh; ++h) ccnt += mask[h + (size_t)H * lane];
    if (__ballot(cnt > 1 || ccnt > 1)) {  // S3D:628-634
      if (lane < nh)
        for (int q = 0; q < nd; ++q) W[lane + nh * q] = C[lane + (size_t)H * q];
      const int rc = se3d_munkres_wave(W, nh, nh, nd, lane, assign);
      if (rc != 0 && lane == 0) atomicOr(&D.flags[f], SE3D_FLAG_ASSIGNMENT_FAILED);
    }
    FF_STAMP(5)   // scans + Hungarian
    bool sp = false;
    const int det = (lane < nh) ? assign : -1;
    if (det >= 0) {
      if (mask[lane + (size_t)H * det]) {
        assoc[(size_t)lane * N + b] = det;
        nviews[lane] += 1;
      } else {
        sp = true;
      }
    }
    bool handled = false;
    for (int h = 0; h < nh; ++h) handled |= (__builtin_amdgcn_readlane(det, h) == lane);
    const unsigned long long lt = (1ull << lane) - 1ull;
    const unsigned long long spb = __ballot(sp);
    const int tot1 = __popcll(spb);
    if (sp) {
      const int nhh = nh + __popcll(spb & lt);
      if (nhh < H) { assoc[(size_t)nhh * N + b] = det; nviews[nhh] = 1; }
    }
    const bool un = (lane < nd) && !handled;  // S3D:662-673
    const unsigned long long unb = __ballot(un);
    const int tot2 = __popcll(unb);
    if (un) {
      const int nhh = nh + tot1 + __popcll(unb & lt);
      if (nhh < H) { assoc[(size_t)nhh * N + b] = lane; nviews[nhh] = 1; }
    }
    const int nn = nh + tot1 + tot2;
    if (nn > H && lane == 0) atomicOr(&D.flags[f], SE3D_FLAG_HYP_OVERFLOW);
    nh = min(nn, H);
    __syncthreads();
    FF_STAMP(6)   // attach / spawn
  }
  for (int i = lane; i < H * N; i += 64) D.assoc[(size_t)f * H * N + i] = assoc[i];
  for (int i = lane; i < H; i += 64) D.nviews[f * H + i] = nviews[i];
  if (lane == 0) D.nhyp[f] = nh;
}
__global__ __launch_bounds__(64) void se3d_k_sweep_wave_lds(Se3dDev D) {
  extern __shared__ __align__(16) double s_sw[];
  const int N = D.d.N, P = D.d.P, H = D.d.H, np = D.d.npairs;
  const int f = blockIdx.x, lane = threadIdx.x;
  SweepWaveLds L;
  L.W = s_sw;
  L.C = L.W + (size_t)H * P;
  L.psc = reinterpret_cast<float *>(L.C + (size_t)H * P);
  L.assoc = reinterpret_cast<int32_t *>(L.psc + (size_t)N * P);
  L.nviews = L.assoc + (size_t)H * N;
  L.nraw = L.nviews + H; L.nvalid = L.nraw + N;
  L.mask = reinterpret_cast<uint8_t *>(L.nvalid + N);
  L.inv = reinterpret_cast<double *>((reinterpret_cast<uintptr_t>(L.mask + (size_t)H * P) + 15) & ~(uintptr_t)15);
  for (int i = lane; i < 2 * N + 1; i += 64) L.inv[i] = i ? 1.0 / (double)i : 0.0;
  for (int i = lane; i < N * P; i += 64) L.psc[i] = D.pscore[(size_t)f * N * P + i];
  for (int i = lane; i < H * N; i += 64) L.assoc[i] = -1;
  for (int i = lane; i < H; i += 64) L.nviews[i] = 0;
  for (int i = lane; i < N; i += 64) { L.nraw[i] = D.nraw[f * N + i]; L.nvalid[i] = D.nvalid[f * N + i]; }
  __syncthreads();
  const double *g = D.pair + (size_t)f * np * P * P;
  // 24-bit multiplies: the launcher admits this kernel for N, P <= 64 only (pairs * P * P <= 2016 * 4096 < 2^24); the 64-bit index arithmetic
  // of the generic expression was four quarter-rate multiplies per table entry on the wavefront's critical path
  se3d_sweep_wave_run(D, f, lane, L, [=](int a, int k, int b, int q, int) { return g[__mul24(__mul24(se3d_pair_index(a, b, N), P) + k, P) + q]; });
}
static size_t se3d_sweep_wave_lds_bytes(int N, int P, int H) {
  return (size_t)H * P * 16 + (size_t)N * P * 4 + (size_t)H * N * 4 + (size_t)H * 4 + (size_t)N * 8 + (((size_t)H * P + 15) & ~(size_t)15) + 16 + (size_t)(2 * N + 1) * 8;
}

// Self-test of se3d_sqrt_rcp_core / se3d_pair_joint_dist against the IEEE expressions on the device (se3d_selftest_exact_math).
// out[0]: square-root and reciprocal-of-the-root mismatches over every float in [2^-60, 2^60]; out[1]: mismatches of n / sqrtf(s) over n_div
// pseudo-random operand pairs.
__global__ __launch_bounds__(256) void se3d_k_selftest_exact_math(unsigned long long *out, unsigned long long n_div) {
  const unsigned long long tid = (unsigned long long)blockIdx.x * 256 + threadIdx.x, nthr = (unsigned long long)gridDim.x * 256;
  unsigned long long bad_s = 0, bad_d = 0;
  for (unsigned long long u = SE3D_EXACT_LO + tid; u <= SE3D_EXACT_HI; u += nthr) {
    const float x = __uint_as_float((unsigned)u);
    float y;
    const float r = se3d_sqrt_rcp_core(x, y);
    bad_s += __float_as_uint(r) != __float_as_uint(sqrtf(x));
    bad_s += !se3d_all_ones(r) && __float_as_uint(y) != __float_as_uint(1.f / r);   // the shared reciprocal RN(1 / root)
  }
  const unsigned span = SE3D_EXACT_HI - SE3D_EXACT_LO + 1u;
  for (unsigned long long i = tid; i < n_div; i += nthr) {
    unsigned long long z = (i + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;   // splitmix64
    z ^= z >> 30; z *= 0x94D049BB133111EBull; z ^= z >> 27; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 31;
    // the distance n / sqrtf(s) as the pair cost forms it: s anywhere in the range (its root is the divisor), the numerator zero every 64th
    // pair, within 2^12 of the root's magnitude on every third pair (quotients around the binade boundary), else anywhere in the range
    const float sq = __uint_as_float(SE3D_EXACT_LO + (unsigned)(z >> 32) % span);
    const float b = sqrtf(sq);
    float a = (i % 3ull == 0ull) ? __uint_as_float((__float_as_uint(b) & 0x7f800000u) - (12u << 23) + (unsigned)z % (24u << 23))
                                 : __uint_as_float(SE3D_EXACT_LO + (unsigned)(z & 0xffffffffull) % span);
    if ((i & 63ull) == 63ull) a = 0.f;
    bad_d += __float_as_uint(se3d_pair_joint_dist(a, sq, 0.f, 1.f)) != __float_as_uint(a / b + 0.f / 1.f);
  }
  if (bad_s) atomicAdd(&out[0], bad_s);
  if (bad_d) atomicAdd(&out[1], bad_d);
}
extern "C" hipError_t se3d_launch_selftest_exact_math(unsigned long long *d_out, unsigned long long n_div, hipStream_t st) {
  hipLaunchKernelGGL(se3d_k_selftest_exact_math, dim3(256 * 64), dim3(256), 0, st, d_out, n_div);
  return hipGetLastError();
}

// Self-test of the rows kernel's quotient (se3d_div_shared on the shared reciprocal se3d_safe_rcp) against the device's IEEE
// quotient (se3d_selftest_shared_quotient).  Operands as the kernel has them under its range guard: divisors = line norms in [2^-30, 2^30]
// whose significand is not all ones (others give y_h = 0 and take the IEEE path: counted in out[1] as skipped), numerators zero (every 64th
// pair) or anywhere in [2^-83, 2^60] — a third of them within 2^12 of the divisor's magnitude, where quotients near 1 exercise the
// binade boundary.  out[0]: mismatches, out[1]: pairs skipped.
__global__ __launch_bounds__(256) void se3d_k_selftest_shared_quotient(unsigned long long *out, unsigned long long n) {
  const unsigned long long tid = (unsigned long long)blockIdx.x * 256 + threadIdx.x, nthr = (unsigned long long)gridDim.x * 256;
  unsigned long long bad = 0, skipped = 0;
  for (unsigned long long i = tid; i < n; i += nthr) {
    unsigned long long z = (i + 0x9E3779B97F4A7C15ull) * 0xBF58476D1CE4E5B9ull;   // splitmix64
    z ^= z >> 30; z *= 0x94D049BB133111EBull; z ^= z >> 27; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 31;
    const unsigned zb = (unsigned)(z >> 32), za = (unsigned)z;
    const float b = __uint_as_float(0x30800000u + zb % (0x4e800000u - 0x30800000u + 1u));          // [2^-30, 2^30]
    float a;
    if (i % 3ull == 0ull) a = __uint_as_float((__float_as_uint(b) & 0x7f800000u) - (12u << 23) + za % (24u << 23));   // within 2^-12 .. 2^12 of b's binade
    else a = __uint_as_float(0x16000000u + za % (0x5d800000u - 0x16000000u + 1u));                   // [2^-83, 2^60]
    if ((i & 63ull) == 63ull) a = 0.f;
    const float y = se3d_safe_rcp(b);
    if (y == 0.f) { ++skipped; continue; }
    bad += __float_as_uint(se3d_div_shared(a, b, y)) != __float_as_uint(a / b);
  }
  if (bad) atomicAdd(&out[0], bad);
  if (skipped) atomicAdd(&out[1], skipped);
}
extern "C" hipError_t se3d_launch_selftest_shared_quotient(unsigned long long *d_out, unsigned long long n, hipStream_t st) {
  hipLaunchKernelGGL(se3d_k_selftest_shared_quotient, dim3(256 * 64), dim3(256), 0, st, d_out, n);
  return hipGetLastError();
}

// Diagnostic (se3d_selftest_dlt32): the float32 DLT solve of the second pass — triangulate() + calcReprojectionError() of the reference,
// S3D:425-465 — on caller-provided systems, one system per lane; compared bit for bit with the CPU restatement in tests/test_gpu_exact32.py.
__global__ __launch_bounds__(64) void se3d_k_selftest_dlt32(int n_sys, int n, const float *__restrict__ Rt, const float *__restrict__ pts, int weighted,
                                                            float *__restrict__ ws, float *__restrict__ Xout, double *__restrict__ eout) {
  const int lane = threadIdx.x, sidx = blockIdx.x * 64 + lane;
  if (sidx >= n_sys) return;
  const X32Mat A{ws + (size_t)blockIdx.x * 64 * 8 * n + lane, 64, 2 * n};
  const float *M = Rt + (size_t)sidx * n * 12, *P = pts + (size_t)sidx * n * 3;
  float amax = 0.f;
  for (int v = 0; v < n; ++v) {
    float r0[4], r1[4];
    x32_dlt_rows(M + 12 * v, P[3 * v], P[3 * v + 1], r0, r1);
    if (weighted) {
      const float cf = P[3 * v + 2];
#pragma unroll
      for (int k = 0; k < 4; ++k) { r0[k] = r0[k] * cf; r1[k] = r1[k] * cf; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) { A.set(2 * v, k, r0[k]); A.set(2 * v + 1, k, r1[k]); amax = fmaxf(amax, fmaxf(fabsf(r0[k]), fabsf(r1[k]))); }
  }
  const float4 v4 = x32_null_vector(2 * n, A.p, A.stride, A.ld, amax);
  const float X[3] = {v4.x / v4.w, v4.y / v4.w, v4.z / v4.w};
  double averageError = 0., norm = 0.;
  for (int v = 0; v < n; ++v) {
    const float *Mv = M + 12 * v;
    const float hx = (Mv[0] * X[0] + (Mv[1] * X[1] + Mv[2] * X[2])) + Mv[3];
    const float hy = (Mv[4] * X[0] + (Mv[5] * X[1] + Mv[6] * X[2])) + Mv[7];
    const float hz = (Mv[8] * X[0] + (Mv[9] * X[1] + Mv[10] * X[2])) + Mv[11];
    const float ix = hx / hz, iy = hy / hz;
    const float dx = ix - P[3 * v], dy = iy - P[3 * v + 1];
    const float error = sqrtf(dx * dx + dy * dy);
    averageError += (double)(P[3 * v + 2] * error);
    norm += (double)P[3 * v + 2];
  }
  Xout[3 * sidx] = X[0]; Xout[3 * sidx + 1] = X[1]; Xout[3 * sidx + 2] = X[2];
  eout[sidx] = averageError / norm;
}
extern "C" hipError_t se3d_launch_selftest_dlt32(int n_sys, int n, const float *d_Rt, const float *d_pts, int weighted, float *d_ws, float *d_X, double *d_err, hipStream_t st) {
  hipLaunchKernelGGL(se3d_k_selftest_dlt32, dim3((n_sys + 63) / 64), dim3(64), 0, st, n_sys, n, d_Rt, d_pts, weighted, d_ws, d_X, d_err);
  return hipGetLastError();
}

__global__ __launch_bounds__(256) void se3d_k_munkres(double *dist, int *assign, int *status, int nR, int nC) {
  __shared__ MunkresShared S;
  __shared__ int a[SE3D_MUNKRES_MAX];
  for (int i = threadIdx.x; i < SE3D_MUNKRES_MAX; i += 256) a[i] = -1;
  __syncthreads();
  const int rc = se3d_munkres_block(dist, nR, nR, nC, S, a);
  for (int r = threadIdx.x; r < nR; r += 256) assign[r] = a[r];
  if (threadIdx.x == 0) *status = rc;
}
__global__ __launch_bounds__(64) void se3d_k_munkres_wave(double *dist, int *assign, int *status, int nR, int nC) {
  int a;
  const int rc = se3d_munkres_wave(dist, nR, nR, nC, threadIdx.x, a);
  if ((int)threadIdx.x < nR) assign[threadIdx.x] = a;
  if (threadIdx.x == 0) *status = rc;
}
// variant: 0 = automatic (single-wavefront code for problems up to 64 x 64), 1 = block-cooperative, 2 = single wavefront
extern "C" void se3d_launch_munkres(double *d_dist, int *d_assign, int *d_status, int nR, int nC, hipStream_t st, int variant) {
  const bool small = nR <= 64 && nC <= 64;
  if (small && variant != 1) hipLaunchKernelGGL(se3d_k_munkres_wave, dim3(1), dim3(64), 0, st, d_dist, d_assign, d_status, nR, nC);
  else hipLaunchKernelGGL(se3d_k_munkres, dim3(1), dim3(256), 0, st, d_dist, d_assign, d_status, nR, nC);
}

// =============================================================================================================
// host-side launch sequence
// =============================================================================================================
// Kernel attributes are per DEVICE in HIP: called by se3d_create for the context's device (after hipSetDevice), not once per process.
extern "C" hipError_t se3d_stage1_init_device(void) {
  return hipFuncSetAttribute(reinterpret_cast<const void *>(se3d_k_sweep_assign), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
}
static void se3d_launch_triangulation(const Se3dDev &D, const se3d_options &o, hipStream_t st, hipEvent_t ev_mid = nullptr) {
  const int B = D.d.B, N = D.d.N, H = D.d.H;
  (void)hipMemsetAsync(D.tri_cnt, 0, sizeof(uint32_t), st);   // list of the second pass: filled by the kernel below
  const int waves2 = (int)std::min<size_t>((size_t)D.d.tri_waves, (size_t)B * H * SE3D_NKP);
  const size_t lds = ((size_t)4 * 15 * N + 12 * N) * sizeof(float);
  if (N <= TH_MAXV && o.tri_form != SE3D_TRI_JOINT) {
    // (the grid covers all H hypothesis slots; the workgroups of unused slots exit after one load — capping the grid at 12 / 16 / 24 slots
    // changed the kernel's 7.6 - 7.7 ms per 98 304 frames by less than the run-to-run spread, profiles/r3/README.md)
    hipLaunchKernelGGL(se3d_k_triangulate_hyp, dim3(TH_PARTS * H, B), dim3(64), 0, st, D);
  }
  else hipLaunchKernelGGL(se3d_k_triangulate, dim3((H * SE3D_NKP + 3) / 4, B), dim3(256), lds, st, D);
  if (ev_mid) (void)hipEventRecord(ev_mid, st);
  // second pass: the listed joints with the oracle's restatement of the reference's float32 arithmetic (wavefronts beyond the list length exit at once)
  // small rigs: four joints per wavefront first (phase A and the fp64 sample covariances); what it marks — and everything when options.exact_cov asks for
  // float32 samples on every listed joint — goes through the one-joint kernel
  const bool quad = N <= SE3D_EXACT4_MAX_N && o.exact_cov != SE3D_EXACT_COV_F32;
  if (quad) hipLaunchKernelGGL(se3d_k_triangulate_exact4, dim3(waves2), dim3(64), se3d_exact4_lds_bytes(N), st, D);
  hipLaunchKernelGGL(se3d_k_triangulate_exact, dim3(waves2), dim3(64), se3d_exact_lds_bytes(N), st, D, o.exact_cov == SE3D_EXACT_COV_F32 ? 1 : 0, quad ? 1 : 0);
}

extern "C" void se3d_launch_normalize_keep(const Se3dDev &D, hipStream_t st) {   // normalised points into D.kp, association workspace untouched
  hipLaunchKernelGGL(se3d_k_normalize, dim3(D.d.N, D.d.B), dim3(256), 0, st, D, 3);
}
extern "C" void se3d_launch_pair_block(const Se3dDev &D, const int2 *d_ab_one, int frame, hipStream_t st) {   // se3d_get_pair_costs under column chunks
  hipLaunchKernelGGL(se3d_k_pairs_rows, dim3((D.d.P + RB_ROWS - 1) / RB_ROWS, 1, 1), dim3(RB_NT), 0, st, D, d_ab_one, 0, 0, frame);
}

// phases (bit mask): 1 = normalise + pair-cost table, 8 = camera sweep, 2 = triangulation, 4 = post-processing + duplicate merge
// st_hi / ev_hi: optional high-priority stream (+ two events) for the per-camera sweep launches of large problems: 2 x N tiny dependent
// launches that otherwise queue behind the millions of workgroups of other contexts' pair kernels.
// o: the context's options (include/se3d.h): the kernel FORMS a parity test may force; everything 0 = the library's choice.
extern "C" void se3d_launch_stage1(const Se3dDev &D, const se3d_options &o, const int2 *pair_ab, hipStream_t st, hipEvent_t *ev /* [7] or null; [6] sits between the two passes of the triangulation */, int phases,
                                   hipStream_t st_hi, hipEvent_t *ev_hi, const Se3dPairPlan *plan, hipEvent_t *ev_chunk) {
  const int B = D.d.B, N = D.d.N, P = D.d.P, H = D.d.H;
  const bool chunked = D.d.pair_chunks > 0 && plan && plan->m == D.d.pair_chunks;
  const bool rows = o.pairs_form == SE3D_PAIRS_ROWS || (o.pairs_form == SE3D_PAIRS_AUTO && P > 16);   // large P: two rows x 32 columns per wavefront
  if ((phases & 1) && chunked) {   // column chunks: phase 1 is the normalisation only, the pair kernel runs chunk by chunk inside phase 8
    if (ev) (void)hipEventRecord(ev[0], st);
    hipLaunchKernelGGL(se3d_k_normalize, dim3(N, B), dim3(256), 0, st, D, 1);
    if (ev) (void)hipEventRecord(ev[1], st);
  } else if (phases & 1) {
    // Small batches (a single frame per call is the reference's operating point, S3D:1069): the fused kernel is ONE workgroup per frame,
    // which leaves the chip empty — the stand-alone normalisation + the flat pair kernel (one lane per table entry over the whole
    // chip) write the same table (tests: flat == fused == rows bitwise) in 14 us instead of 47 us at B = 1 (16cam6p).  Up to 65 536 table
    // entries per launch (8 frames of 16cam6p).  Rigs with N * P > 256 and P <= 16 take the flat kernel as well.
    const bool small_batch = o.pairs_form == SE3D_PAIRS_AUTO && !rows && (size_t)B * D.d.npairs * P * P <= ((size_t)1 << 16);
    const bool fused_np = !rows && o.pairs_form != SE3D_PAIRS_FLAT && !small_batch && N * P <= NP_MAX_DETS && D.d.npairs <= 2048;
    if (ev) (void)hipEventRecord(ev[0], st);
    if (!fused_np) hipLaunchKernelGGL(se3d_k_normalize, dim3(N, B), dim3(256), 0, st, D, 1);
    if (ev) (void)hipEventRecord(ev[1], st);
    if (fused_np) {  // normalisation + pair table in one workgroup per frame, normalised points only in LDS
      const size_t lds = (size_t)N * P * SE3D_NKP * sizeof(float2) + (size_t)N * P * 8 + (size_t)N * 4 + (size_t)(D.d.npairs + 1) * 4;
      hipLaunchKernelGGL(se3d_k_normalize_pairs, dim3(B), dim3(256), lds, st, D, pair_ab);
    } else if (rows) {
      hipLaunchKernelGGL(se3d_k_pairs_rows, dim3((P + RB_ROWS - 1) / RB_ROWS, D.d.npairs, B), dim3(RB_NT), 0, st, D, pair_ab, 0, 0, 0);
    } else {   // the reference form: one lane per (frame, pair, det_a, det_b) entry, IEEE divide
      const size_t total = (size_t)B * D.d.npairs * P * P;
      const int blocks = (int)((total + 255) / 256 < 65536 ? (total + 255) / 256 : 65536);
      hipLaunchKernelGGL(se3d_k_pairs_flat, dim3(blocks), dim3(256), 0, st, D, pair_ab);
    }
    if (ev) (void)hipEventRecord(ev[2], st);
  }
  if ((phases & 8) && chunked) {
    // Pair kernel and camera sweep interleaved, chunk by chunk: chunk j of the pair table (columns col0[j] .. col0[j + 1] - 1) goes to
    // buffer j % 2 of the frame's pair slots on the bulk stream while the sweep of chunk j - 1 reads the other buffer on the
    // (high-priority) sweep stream; chunk j + 2 waits until the sweep has left buffer j % 2.
    // Munkres working matrix: in L2 here (dynW = 0).  In LDS (up to 128 KB per frame's workgroup, options.sweep_lds_kb = 128) a resident sweep
    // workgroup leaves room for one pair-kernel workgroup on its CU instead of four; with 64 frames per context in flight that cost more
    // than the faster Munkres gave back (26.6 k vs 27.6 k skeletons/s at 256cam128p, profiles/r3/README.md).
    const int dynW = o.sweep_lds_kb > 0 ? std::min((int)std::min((size_t)H * P * sizeof(double), (size_t)128 * 1024), o.sweep_lds_kb * 1024) : 0;
    hipStream_t sw = (st_hi && ev_chunk) ? st_hi : st;
    const int m = plan->m, cap = plan->cap;
    hipEvent_t *ev_p = ev_chunk, *ev_s = ev_chunk ? ev_chunk + SE3D_PAIR_CHUNKS_MAX : nullptr;
    for (int j = 0; j < m; ++j) {
      const int n_pairs = plan->list0[j + 1] - plan->list0[j], slot0 = (j & 1) * cap;
      if (sw != st && j >= 2) (void)hipStreamWaitEvent(st, ev_s[j - 2], 0);
      hipLaunchKernelGGL(se3d_k_pairs_rows, dim3((P + RB_ROWS - 1) / RB_ROWS, n_pairs, B), dim3(RB_NT), 0, st, D, pair_ab, plan->list0[j], slot0, 0);
      if (sw != st) {
        (void)hipEventRecord(ev_p[j], st);
        (void)hipStreamWaitEvent(sw, ev_p[j], 0);
      }
      if (j == 0) hipLaunchKernelGGL(se3d_k_sweep_assign, dim3(B), dim3(256), (size_t)dynW, sw, D, 0, dynW);   // camera 0 seeds the hypotheses
      for (int b = plan->col0[j]; b < plan->col0[j + 1]; ++b) {
        hipLaunchKernelGGL(se3d_k_sweep_cost, dim3(H, B), dim3(128), 0, sw, D, b, slot0 - plan->list0[j]);
        hipLaunchKernelGGL(se3d_k_sweep_assign, dim3(B), dim3(256), (size_t)dynW, sw, D, b, dynW);
      }
      if (sw != st) (void)hipEventRecord(ev_s[j], sw);
    }
    if (ev) (void)hipEventRecord(ev[2], st);   // "pairs": the bulk stream up to its last chunk (waits for free buffers included)
    if (sw != st) (void)hipStreamWaitEvent(st, ev_s[m - 1], 0);
  } else if (phases & 8) {
    // Three launch structures with bytewise equal association tables (tests/test_gpu_stage1.py): one wavefront per frame with the frame's
    // state in LDS (small rigs), one wavefront per frame with the state in global memory (mid-size problems), one cost + one assignment
    // launch per camera (everything else: one wavefront per frame is too little parallelism there).
    // Munkres working matrix of the per-camera form in LDS when H * P doubles fit next to the static LDS of the kernel (<= 128 KB;
    // the dynamic-LDS limit of se3d_k_sweep_assign is raised per device by se3d_stage1_init_device).
    int dynW = (int)((size_t)H * P * sizeof(double));
    if (dynW > 128 * 1024) dynW = 128 * 1024;
    const bool wave_lds_ok = H <= 64 && P <= 64 && N <= 64;   // (the LDS form holds per-camera words for N <= 64 cameras)
    if (o.sweep_form == SE3D_SWEEP_AUTO && wave_lds_ok && se3d_sweep_wave_lds_bytes(N, P, H) <= 16 * 1024) {
      hipLaunchKernelGGL(se3d_k_sweep_wave_lds, dim3(B), dim3(64), se3d_sweep_wave_lds_bytes(N, P, H), st, D);
    } else if (o.sweep_form != SE3D_SWEEP_PER_CAMERA && se3d_sweep_wave_global_ok(P, H)) {   // any N (N > 64 with a small H x P lost this form in round 5)
      hipLaunchKernelGGL(se3d_k_sweep_fused_wave, dim3(B), dim3(64), (size_t)dynW, st, D);
    } else {
      hipStream_t sw = st;
      if (st_hi && ev_hi) {  // pair table complete -> the sweep chain runs on the high-priority stream -> triangulation waits for it
        (void)hipEventRecord(ev_hi[0], st);
        (void)hipStreamWaitEvent(st_hi, ev_hi[0], 0);
        sw = st_hi;
      }
      for (int b = 0; b < N; ++b) {
        if (b > 0) hipLaunchKernelGGL(se3d_k_sweep_cost, dim3(H, B), dim3(128), 0, sw, D, b, 0);
        hipLaunchKernelGGL(se3d_k_sweep_assign, dim3(B), dim3(256), (size_t)dynW, sw, D, b, dynW);
      }
      if (sw != st) {
        (void)hipEventRecord(ev_hi[1], sw);
        (void)hipStreamWaitEvent(st, ev_hi[1], 0);
      }
    }
  }
  if (ev && (phases & 8)) (void)hipEventRecord(ev[3], st);
  if (phases & 2) se3d_launch_triangulation(D, o, st, ev ? ev[6] : nullptr);
  if (ev && (phases & 2)) (void)hipEventRecord(ev[4], st);
  if (phases & 4) {
    if (H <= SE3D_POST_SMALL_H && o.post_form != SE3D_POST_BLOCK) hipLaunchKernelGGL(se3d_k_postprocess_wave3, dim3(B), dim3(64), se3d_post_wave3_lds_bytes(H), st, D);
    else hipLaunchKernelGGL(se3d_k_postprocess, dim3(B), dim3(256), 0, st, D);
  }
  if (ev && (phases & 4)) (void)hipEventRecord(ev[5], st);
}

// =============================================================================================================
// Hypothesis shard (north_star config 5): the association is replicated on every rank, rank r triangulates the hypotheses
// h with h % world == r, the ranks all-gather their [frame][ceil(H / world)] blocks of hypothesis persons and every rank scatters all
// blocks back into its hypothesis slots before the (replicated) post-processing and duplicate merge.
// =============================================================================================================
__global__ __launch_bounds__(64) void se3d_k_shard_pack_hyp(Se3dDev D, se3d_person3d *__restrict__ out) {
  const int j = blockIdx.x, f = blockIdx.y, W = D.d.shard_world, Hs = gridDim.x, h = j * W + D.d.shard_rank;
  constexpr int NW = (int)(sizeof(se3d_person3d) / 4);
  uint32_t *dst = reinterpret_cast<uint32_t *>(out + (size_t)f * Hs + j);
  if (h < D.d.H) {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(D.hypP + (size_t)f * D.d.H + h);
    for (int i = threadIdx.x; i < NW; i += 64) dst[i] = src[i];
  } else {
    for (int i = threadIdx.x; i < NW; i += 64) dst[i] = 0u;
  }
}
__global__ __launch_bounds__(64) void se3d_k_shard_unpack_hyp(Se3dDev D, const se3d_person3d *__restrict__ gathered, int Hs) {
  const int W = D.d.shard_world, r = blockIdx.x / Hs, j = blockIdx.x - r * Hs, f = blockIdx.y, h = j * W + r;
  if (h >= D.d.H || r == D.d.shard_rank) return;   // own hypotheses are already in place
  constexpr int NW = (int)(sizeof(se3d_person3d) / 4);
  const uint32_t *src = reinterpret_cast<const uint32_t *>(gathered + ((size_t)r * D.d.B + f) * Hs + j);
  uint32_t *dst = reinterpret_cast<uint32_t *>(D.hypP + (size_t)f * D.d.H + h);
  for (int i = threadIdx.x; i < NW; i += 64) dst[i] = src[i];
}
extern "C" void se3d_launch_shard_pack_hyp(const Se3dDev &D, se3d_person3d *out, hipStream_t st) {
  const int Hs = (D.d.H + D.d.shard_world - 1) / D.d.shard_world;
  hipLaunchKernelGGL(se3d_k_shard_pack_hyp, dim3(Hs, D.d.B), dim3(64), 0, st, D, out);
}
extern "C" void se3d_launch_shard_unpack_hyp(const Se3dDev &D, const se3d_person3d *gathered, hipStream_t st) {
  const int Hs = (D.d.H + D.d.shard_world - 1) / D.d.shard_world;
  hipLaunchKernelGGL(se3d_k_shard_unpack_hyp, dim3(Hs * D.d.shard_world, D.d.B), dim3(64), 0, st, D, gathered, Hs);
}

// =============================================================================================================
// Config 2 of SURVEY 8d — batched DLT with the association GIVEN (se3d_triangulate_assoc): rows N, T, P, O, U only.
// The caller's table holds ORIGINAL detection indices ([frame][hypothesis][camera], the layout of se3d_frame_out.hyp_assoc); row N
// still applies its ">= 9 valid keypoints" filter (S3D:579, 599), so a listed detection that the sweep could never have attached maps
// to "no view".  One workgroup per frame.
// =============================================================================================================
__global__ __launch_bounds__(256) void se3d_k_assoc_given(Se3dDev D, const int32_t *__restrict__ assoc_orig, int nh) {
  const int f = blockIdx.x, tid = threadIdx.x;
  const int N = D.d.N, P = D.d.P, H = D.d.H;
  for (int i = tid; i < nh * N; i += 256) {
    const int h = i / N, c = i - h * N;
    const int o = assoc_orig[((size_t)f * nh + h) * N + c];
    int slot = -1;
    if (o >= 0) {
      const size_t base = ((size_t)(f * N + c)) * P;
      const int nv = D.nvalid[f * N + c];
      for (int k = 0; k < nv; ++k)
        if (D.orig[base + k] == o) slot = k;
    }
    D.assoc[((size_t)f * H + h) * N + c] = slot;
  }
  __syncthreads();
  for (int h = tid; h < nh; h += 256) {
    int n = 0;
    for (int c = 0; c < N; ++c) n += D.assoc[((size_t)f * H + h) * N + c] >= 0 ? 1 : 0;
    D.nviews[f * H + h] = n;
  }
  if (tid == 0) D.nhyp[f] = nh;
  __syncthreads();
  for (int h = 0; h < nh; ++h) {  // hypotheses the triangulation skips (< 2 views, S3D:652): default-constructed persons
    if (D.nviews[f * H + h] >= 2) continue;
    uint32_t *w = reinterpret_cast<uint32_t *>(&D.hypP[(size_t)f * H + h]);
    for (int i = tid; i < (int)(sizeof(se3d_person3d) / 4); i += 256) w[i] = 0u;
  }
}

extern "C" void se3d_launch_dlt(const Se3dDev &D, const se3d_options &o, const int32_t *d_assoc_orig, int nh, hipStream_t st, hipEvent_t *ev /* [2] or null */) {
  const int B = D.d.B, N = D.d.N;
  if (ev) (void)hipEventRecord(ev[0], st);
  hipLaunchKernelGGL(se3d_k_normalize, dim3(N, B), dim3(256), 0, st, D, 0);
  hipLaunchKernelGGL(se3d_k_assoc_given, dim3(B), dim3(256), 0, st, D, d_assoc_orig, nh);
  se3d_launch_triangulation(D, o, st);
  if (ev) (void)hipEventRecord(ev[1], st);
}
