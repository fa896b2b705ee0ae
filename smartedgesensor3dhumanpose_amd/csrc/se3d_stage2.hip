// se3d_stage2.hip — pose_prior refinement on CDNA4 (gfx950): track association (PP:548-580), per-person factor graph
// (unary Gaussian + bone-length range factors, PP:626-744) optimised with Levenberg-Marquardt under GTSAM 4.0.3 default
// semantics (PP:746-758), marginal covariances (PP:760-793), de-normalisation + velocity prediction (PP:769-852) and
// track lifecycle (PP:191-211, 867-903).  SURVEY §8a rows Y, A2, F, R, Z, L.
//
// One wavefront per person: the whitened Jacobian J (<= 84 x 63) is never materialised — MFMA operands for the dense
// J^T J (v_mfma_f64_16x16x4_f64, the only MFMA use on the path) are generated on the fly from the compact factor
// data; H = J^T J + lambda I lives in LDS (64 x 65 doubles), Cholesky / triangular solves / inverse run on the 64 lanes.
#include "se3d_device.h"
#include "se3d_munkres.h"

#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <type_traits>

// Stage 2 is fp64 optimisation with a stated tolerance (refined joints < 1 mm, GTSAM 4.0.3 control flow), not a bit-exact float
// expression tree like stage 1: let the compiler contract a*b+c into v_fma_f64 here (shorter dependency chains in the LM solves).
// The Munkres code included above only adds, subtracts and compares, so its bit-exactness is unaffected.
#pragma clang fp contract(fast)

typedef double d4 __attribute__((ext_vector_type(4)));

// Diagnostic build only (make diag): per-phase cycle stamps of the LM kernel written to a debug buffer (never in the product .so)
#ifdef SE3D_LM_STAMPS
__device__ unsigned long long g_lm_stamps[16];
#define LM_STAMP(idx)                                                                 \
  {                                                                                   \
    const unsigned long long now_ = __builtin_readcyclecounter();                     \
    if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) atomicAdd(&g_lm_stamps[idx], now_ - stamp_prev_);   /* one workgroup in 64: the counters are one cache line */ \
    stamp_prev_ = __builtin_readcyclecounter();                                       \
  }
#define LM_STAMP_INIT unsigned long long stamp_prev_ = __builtin_readcyclecounter();
__device__ int g_lm_ablate;  // diagnostic build only: 1 = stop after one LM iteration, 2 = skip marginals, 4 = skip the LM loop,
                             // (with 1:) 8 = skip the solve, 16 = skip linearize, 32 = skip the linearised error, 64 = skip the new error
#define LM_ABLATE(bit) (g_lm_ablate & (bit))
#else
#define LM_ABLATE(bit) 0
#define LM_STAMP(idx)
#define LM_STAMP_INIT
#endif

__device__ __constant__ const double c_vel_sigmas[SE3D_NFU] = {2., 1., 1., 2., 3., 1., 2., 3., 1., 1., 2., 3., 1., 2., 3., 2., 2., 2., 2., 2., 1.};  // FBP:33
struct Se3dBone { int k1, k2; double L, sigma; int not_belly; };
// addBinaryFactors: [0] absolute bone lengths (PP:434-479), [1] height-normalised (PP:386-431)
__device__ __constant__ const Se3dBone c_bones[2][21] = {
    {{8, 9, 0.134, 0.033, 0},   {8, 12, 0.134, 0.033, 0},  {9, 10, 0.449, 0.051, 0},  {10, 11, 0.446, 0.051, 0}, {12, 13, 0.449, 0.051, 0},
     {13, 14, 0.446, 0.051, 0}, {1, 0, 0.20, 0.025, 0},    {1, 2, 0.15, 0.042, 0},    {1, 5, 0.15, 0.042, 0},    {2, 3, 0.28, 0.045, 0},
     {3, 4, 0.25, 0.063, 0},    {5, 6, 0.28, 0.045, 0},    {6, 7, 0.25, 0.063, 0},    {8, 20, 0.23846, 0.071, 0}, {20, 1, 0.25534, 0.035, 0},
     {0, 19, 0.11500, 0.035, 0}, {8, 1, 0.50, 0.071, 1},   {0, 15, 0.05, 0.035, 0},   {0, 16, 0.05, 0.035, 0},   {15, 17, 0.10, 0.05, 0},
     {16, 18, 0.10, 0.05, 0}},
    {{8, 9, 0.17, 0.062, 0},    {8, 12, 0.17, 0.062, 0},   {9, 10, 0.694, 0.111, 0},  {10, 11, 0.708, 0.097, 0}, {12, 13, 0.694, 0.111, 0},
     {13, 14, 0.708, 0.097, 0}, {1, 0, 0.33, 0.050, 0},    {1, 2, 0.262, 0.092, 0},   {1, 5, 0.262, 0.092, 0},   {2, 3, 0.515, 0.071, 0},
     {3, 4, 0.444, 0.084, 0},   {5, 6, 0.515, 0.071, 0},   {6, 7, 0.444, 0.084, 0},   {8, 20, 0.49, 0.05, 0},    {20, 1, 0.51, 0.05, 0},
     {0, 19, 0.23, 0.05, 0},    {8, 1, 1.000, 0.02, 1},    {0, 15, 0.085, 0.06, 0},   {0, 16, 0.085, 0.06, 0},   {15, 17, 0.167, 0.08, 0},
     {16, 18, 0.167, 0.08, 0}}};

__device__ __forceinline__ int s2_block_excl_scan(bool flag, int *wsum, int &total) {
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

// remove_old_tracks (PP:191-211): stable erase of tracks unobserved for more than 1 s.  blockDim = 256 >= T.
__device__ void s2_remove_old(const Se3dStage2 &Q, int s, double t, int *wsum, int *tmp /*LDS[256]*/) {
  const int tid = threadIdx.x, T = Q.T;
  const int nt = Q.n_tracks[s];
  int *order = Q.order + (size_t)s * T;
  int slot = -1;
  bool keep = false;
  if (tid < nt) {
    slot = order[tid];
    keep = !(t - Q.trk_t_prev[(size_t)s * T + slot] > 1.0);
    if (!keep) Q.used[(size_t)s * T + slot] = 0;
  }
  int total;
  const int pos = s2_block_excl_scan(keep, wsum, total);
  if (keep) tmp[pos] = slot;
  __syncthreads();
  if (tid < total) order[tid] = tmp[tid];
  if (tid == 0) Q.n_tracks[s] = total;
  __syncthreads();
}

// =============================================================================================================
// row Y — calc_normed_dist (PP:84-101) for every (detection, track): one workgroup per (track, sequence), threads over
// detections; the track's previous estimate is de-normalised once into LDS.  C is column-major [det + n_det * track].
// =============================================================================================================
__device__ __forceinline__ void s2_track_cost_body(const Se3dStage2 &Q, int tk, int s, int tid, int nthr, double (*pp)[3], double *wgt) {
  const int T = Q.T, H = Q.H;
  const int nd = min(Q.n_det[s], H);
  const size_t sl = (size_t)s * T + Q.order[(size_t)s * T + tk];
  const uint32_t has = Q.has[sl];
  const double delta_t = Q.t[s] - Q.trk_t_prev[sl];
  if (tid < SE3D_NFU * 3) {
    const int k = tid / 3, a = tid - 3 * k;
    pp[k][a] = Q.prev[(sl * SE3D_NFU + k) * 3 + a] * Q.height_prev[sl] + Q.root_prev[sl * 3 + a];
    if (a == 0) wgt[k] = c_vel_sigmas[k] * delta_t;
  }
  __syncthreads();
  const se3d_person3d *dets = Q.dets + (size_t)s * H;
  double *C = Q.C + (size_t)s * H * T;
  for (int det = tid; det < nd; det += nthr) {
    int nj = 0;
    double dist = 0.0;
    for (int k0 = 0; k0 < SE3D_NFU; k0 += 7) {  // seven keypoints in flight per memory round trip (a load per joint behind its own test: 21 trips)
      double jx[7], jy[7], jz[7];
      float sc[7];
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        const se3d_keypoint3d &kp = dets[det].keypoints[k0 + q];
        jx[q] = kp.joint[0]; jy[q] = kp.joint[1]; jz[q] = kp.joint[2]; sc[q] = kp.score;
      }
#pragma unroll
      for (int q = 0; q < 7; ++q) {
        const int k = k0 + q;
        if (sc[q] > 0.10f && ((has >> k) & 1u)) {
          const double dx = jx[q] - pp[k][0], dy = jy[q] - pp[k][1], dz = jz[q] - pp[k][2];
          dist += sqrt(dx * dx + dy * dy + dz * dz) / wgt[k];
          ++nj;
        }
      }
    }
    C[det + (size_t)nd * tk] = nj > 0 ? dist / nj : 1e6;
  }
}
__global__ __launch_bounds__(128) void se3d_k_track_cost(Se3dStage2 Q) {
  const int tk = blockIdx.x, s = blockIdx.y + Q.seq0;
  if (min(Q.n_det[s], Q.H) == 0 || tk >= Q.n_tracks[s]) return;
  __shared__ double pp[SE3D_NFU][3];
  __shared__ double wgt[SE3D_NFU];
  s2_track_cost_body(Q, tk, s, threadIdx.x, 128, pp, wgt);
}

// row L (distance part) — calc_3d_dist (PP:103-119) between every pair of live tracks of a sequence, as a bit matrix over
// track slots.  Merging never modifies the surviving track (PP:872-876), so all tests can be evaluated up front.
__device__ __forceinline__ void s2_track_near_body(const Se3dStage2 &Q, int si, int s, int lane, double (*pi)[3]) {  // one wavefront
  const int T = Q.T;
  const size_t sli = (size_t)s * T + si;
  unsigned long long *row = Q.near + sli * 4;
  if (!Q.used[sli]) { if (lane < 4) row[lane] = 0ull; return; }
  const uint32_t hasi = Q.has[sli];
  if (lane < SE3D_NFU)
    for (int a = 0; a < 3; ++a) pi[lane][a] = Q.prev[(sli * SE3D_NFU + lane) * 3 + a] * Q.height_prev[sli] + Q.root_prev[sli * 3 + a];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  for (int w = 0; w < 4; ++w) {
    const int sj = w * 64 + lane;
    bool nr = false;
    if (sj < T && sj != si && Q.used[(size_t)s * T + sj]) {
      const size_t slj = (size_t)s * T + sj;
      const uint32_t both = hasi & Q.has[slj];
      const double hj = Q.height_prev[slj];
      int nj = 0;
      double dist = 0.0;
      for (int k = 0; k < SE3D_NFU; ++k)
        if ((both >> k) & 1u) {
          const double dx = pi[k][0] - (Q.prev[(slj * SE3D_NFU + k) * 3 + 0] * hj + Q.root_prev[slj * 3 + 0]);
          const double dy = pi[k][1] - (Q.prev[(slj * SE3D_NFU + k) * 3 + 1] * hj + Q.root_prev[slj * 3 + 1]);
          const double dz = pi[k][2] - (Q.prev[(slj * SE3D_NFU + k) * 3 + 2] * hj + Q.root_prev[slj * 3 + 2]);
          dist += sqrt(dx * dx + dy * dy + dz * dz);
          ++nj;
        }
      nr = (nj > 0 ? dist / nj : 1e6) < 0.20;
    }
    const unsigned long long b = __ballot(nr);
    if (lane == 0) row[w] = b;
  }
}
__global__ __launch_bounds__(64) void se3d_k_track_near(Se3dStage2 Q) {
  const int si = blockIdx.x, s = blockIdx.y + Q.seq0;
  if (min(Q.n_det[s], Q.H) == 0) return;
  __shared__ double pi[SE3D_NFU][3];
  s2_track_near_body(Q, si, s, threadIdx.x, pi);
}

// =============================================================================================================
// rows Y + A2 — track <-> detection association (PP:513-580).  One workgroup per sequence.
// =============================================================================================================
__device__ __forceinline__ void s2_track_assoc_body(const Se3dStage2 &Q, int s, double *s2_dynW, int dyn_lds_bytes) {  // blockDim.x == 256
  const int tid = threadIdx.x;
  const int T = Q.T, H = Q.H, N = Q.N;
  __shared__ MunkresShared MS;
  __shared__ int assign[SE3D_MUNKRES_MAX];
  __shared__ int freelist[SE3D_MUNKRES_MAX];
  __shared__ int wsum[4];
  const double t = Q.t[s];
  const int nd = min(Q.n_det[s], H);
  if (tid == 0) {  // feedback-delay moving average -> pred_delta_t (PP:513-526)
    double acc = 0.0;
    int nv = 0;
    for (int i = 0; i < N; ++i) {
      const float d = Q.fb_delay ? Q.fb_delay[(size_t)s * N + i] : -1.0f;
      if (d > 0.0f) { acc += (double)d; ++nv; }
    }
    const double cur = nv > 0 ? acc / nv : 0.10;
    Q.fb_buf[s * 3 + (Q.frame_nr[s] % 3)] = cur;
    Q.pred_dt[s] = (0.0 + Q.fb_buf[s * 3 + 0] + Q.fb_buf[s * 3 + 1] + Q.fb_buf[s * 3 + 2]) / 3;
    Q.n_out[s] = 0;
  }
  for (int i = tid; i < H; i += 256) { Q.det_slot[(size_t)s * H + i] = -1; Q.pub[(size_t)s * H + i] = 0; }
  for (int i = tid; i < SE3D_MUNKRES_MAX; i += 256) assign[i] = -1;
  __syncthreads();
  if (nd == 0) {  // PP:537-546 (g_frame_nr is not advanced on empty frames)
    s2_remove_old(Q, s, t, wsum, freelist);
    if (tid == 0) Q.t_prev[s] = t;
    return;
  }
  const int nt = Q.n_tracks[s];
  int *order = Q.order + (size_t)s * T;
  const bool w_in_lds = (size_t)nd * nt * sizeof(double) <= (size_t)dyn_lds_bytes;
  double *C = Q.C + (size_t)s * H * T, *Wg = Q.W + (size_t)s * H * T;
  // C from se3d_k_track_cost; Munkres works on a copy (LDS when it fits, else HBM)
  if (w_in_lds) { for (int idx = tid; idx < nd * nt; idx += 256) s2_dynW[idx] = C[idx]; }
  else { for (int idx = tid; idx < nd * nt; idx += 256) Wg[idx] = C[idx]; }
  __syncthreads();
  if (nt > 0) {
    int rc = 0;
    if (w_in_lds && nd <= 64 && nt <= 64) {  // actual problem fits one wavefront (whatever the capacities H, T): register-resident Munkres
      if (tid < 64) {
        int a = -1;
        rc = se3d_munkres_wave(s2_dynW, nd, nd, nt, tid, a);
        if (tid < nd) assign[tid] = a;
      }
      __syncthreads();
    } else if (w_in_lds) rc = se3d_munkres_block(s2_dynW, nd, nd, nt, MS, assign);
    else rc = se3d_munkres_block(Wg, nd, nd, nt, MS, assign);
    if (rc != 0 && tid == 0) atomicOr(&Q.flags2[s], 4u);
    if (tid < nd) {  // PP:562-565
      const int a = assign[tid];
      if (a >= 0 && C[tid + (size_t)nd * a] > 5.0) assign[tid] = -1;
    }
    __syncthreads();
  }
  // new tracks for unassigned detections, in detection order (PP:570-580)
  const bool isnew = (tid < nd) && (nt == 0 || assign[tid] < 0);
  int total_new, total_free;
  const int p = s2_block_excl_scan(isnew, wsum, total_new);
  const bool fr = (tid < T) && !Q.used[(size_t)s * T + tid];
  const int q = s2_block_excl_scan(fr, wsum, total_free);
  if (fr) freelist[q] = tid;
  __syncthreads();
  if (tid < nd) {
    if (!isnew) {
      Q.det_slot[(size_t)s * H + tid] = order[assign[tid]];
    } else if (nt + p < T && p < total_free) {
      const int slot = freelist[p];
      const size_t sl = (size_t)s * T + slot;
      Q.used[sl] = 1;
      Q.id[sl] = Q.next_id[s] + p;
      Q.num_obs[sl] = 0;
      Q.has[sl] = 0;
      Q.trk_t_prev[sl] = t;  // the reference leaves t_prev uninitialised until the first update (PP:79-82)
      Q.height_prev[sl] = -1.0;
      for (int a = 0; a < 3; ++a) Q.root_prev[sl * 3 + a] = 0.0;
      for (int a = 0; a < SE3D_NFU * 9; ++a) Q.vel[sl * SE3D_NFU * 9 + a] = 0.0;
      order[nt + p] = slot;
      Q.det_slot[(size_t)s * H + tid] = slot;
    } else {
      atomicOr(&Q.flags2[s], 1u);  // track capacity exceeded
    }
  }
  __syncthreads();
  if (tid == 0) {
    const int created = min(total_new, min(T - nt, total_free));
    Q.n_tracks[s] = nt + created;
    Q.next_id[s] += total_new;
  }
}

__global__ __launch_bounds__(256) void se3d_k_track_assoc(Se3dStage2 Q, int dyn_lds_bytes) {
  extern __shared__ __align__(16) double s2_dynW_k[];  // Munkres working matrix when it fits, else HBM workspace
  s2_track_assoc_body(Q, blockIdx.x + Q.seq0, s2_dynW_k, dyn_lds_bytes);
}

// =============================================================================================================
// rows F, R, Z — one wavefront per detection: build the graph, LM, marginals, outputs, track-state update.
// =============================================================================================================
#define HS 65  // LDS row stride of H (doubles): column accesses by 64 lanes stay conflict-free

struct LmCommon {
  double Rm[SE3D_NFU][9];   // whitening matrix of each variable's unary factor (R^T R = Sigma^-1)
  double mm[SE3D_NFU][3];   // unary measurement
  double xc[SE3D_NFU][3];   // current values
  double xn[SE3D_NFU][3];   // tentative values
  double bb[88];            // -whitened error (unary rows 3v+i, then range rows)
  double Hs[21][3];         // range Jacobian direction / sigma
  double rL[21], rS[21];
  double dl[64];
  int rv1[21], rv2[21];
  int keyofcol[SE3D_NFU];
};
struct LmLds : LmCommon {   // dense variant: H = J^T J + lambda I (64 x 65 doubles)
  double Hm[64 * HS];
};
struct LmSchurBase : LmCommon {  // what the forest (tree) elimination needs besides the factor data
  double mcov[SE3D_NFU][6];
  unsigned incA[SE3D_NFU], incB[SE3D_NFU];  // bit e: variable is the first / second end of bone factor e
  double al[SE3D_NFU], be[SE3D_NFU];
  int par[SE3D_NFU], lev[SE3D_NFU], pe[SE3D_NFU];
};
// Schur/Woodbury workspace: only the <= 21 x 21 bone-factor system is dense
#define S2_DV_DOUBLES (SE3D_NFU * 6)   // (A_v + lambda I)^-1, symmetric 3x3 (xx,xy,xz,yy,yz,zz)
#define S2_U_DOUBLES (21 * 3)          // D^-1 W columns: -D_{v1}^-1 hs_e (ua) and +D_{v2}^-1 hs_e (ub); D^-1 g (zz)
#define S2_YY_DOUBLES 32
#define S2_MM_DOUBLES (32 * 33)        // M = I + W^T D^-1 W, later its Cholesky factor
#define S2_MI_DOUBLES (21 * 22)        // M^-1 (marginals)
#define S2_SCHUR_SCRATCH_DOUBLES (S2_DV_DOUBLES + 3 * S2_U_DOUBLES + S2_YY_DOUBLES + S2_MM_DOUBLES + S2_MI_DOUBLES)
struct LmSchurLds : LmSchurBase {  // Schur workspace in LDS (per-step kernels)
  double Dv[SE3D_NFU][6];
  double ua[21][3], ub[21][3];
  double zz[SE3D_NFU][3];
  double yy[32];
  double Mm[32][33];
  double Mi[21][22];
};
struct LmTreeLds : LmSchurBase {   // fused walk: the Schur workspace is needed only by the non-forest fallback (never taken with the
  double (*Dv)[6];                 // reference's bone tables) and lives in a per-wavefront HBM scratch: 7.5 KB of LDS per LM wavefront
  double (*ua)[3], (*ub)[3];       // instead of 22.5 KB
  double (*zz)[3];
  double *yy;
  double (*Mm)[33];
  double (*Mi)[22];
  __device__ void bind(double *scr) {
    Dv = reinterpret_cast<double (*)[6]>(scr); scr += S2_DV_DOUBLES;
    ua = reinterpret_cast<double (*)[3]>(scr); scr += S2_U_DOUBLES;
    ub = reinterpret_cast<double (*)[3]>(scr); scr += S2_U_DOUBLES;
    zz = reinterpret_cast<double (*)[3]>(scr); scr += S2_U_DOUBLES;
    yy = scr; scr += S2_YY_DOUBLES;
    Mm = reinterpret_cast<double (*)[33]>(scr); scr += S2_MM_DOUBLES;
    Mi = reinterpret_cast<double (*)[22]>(scr);
  }
};

// Wavefront sum without the LDS crossbar: four DPP steps reduce every 16-lane row in the VALU (xor 1, xor 2, half-mirror, mirror),
// v_readlane collects the four row sums.  ds_bpermute-based butterflies cost ~1.2 k cycles of dependent latency per sum and sit on
// the critical path of every LM trial step (linearised error, non-linear error).
template <int CTRL>
__device__ __forceinline__ double s2_dpp_f64(double v) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, 0xf, 0xf, false);
  return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double s2_readlane_f64(double v, int l) {
  const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)u, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(u >> 32), l);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double s2_wave_sum(double v) {
  v += s2_dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
  v += s2_dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]
  v += s2_dpp_f64<0x141>(v);   // row_half_mirror
  v += s2_dpp_f64<0x140>(v);   // row_mirror
  return (s2_readlane_f64(v, 0) + s2_readlane_f64(v, 16)) + (s2_readlane_f64(v, 32) + s2_readlane_f64(v, 48));
}

// noiseModel::Gaussian::Covariance(cov, smart): |off-diagonal| <= 1e-9 (upper triangle) -> Diagonal sigmas, else
// R = upper Cholesky factor of cov^-1.  Returns false if cov^-1 is not positive definite.
__device__ bool s2_make_whitener(const double cov[9], double R[9]) {
  for (int i = 0; i < 9; ++i) R[i] = 0.0;
  const bool full = fabs(cov[1]) > 1e-9 || fabs(cov[2]) > 1e-9 || fabs(cov[5]) > 1e-9;
  if (!full) {
    R[0] = 1.0 / sqrt(cov[0]); R[4] = 1.0 / sqrt(cov[4]); R[8] = 1.0 / sqrt(cov[8]);
    return true;
  }
  const double a = cov[0], b = cov[1], c = cov[2], d = cov[3], e = cov[4], f = cov[5], g = cov[6], h = cov[7], i9 = cov[8];
  const double c00 = e * i9 - f * h, c01 = -(d * i9 - f * g), c02 = d * h - e * g;
  const double det = a * c00 + b * c01 + c * c02, id = 1.0 / det;
  const double inv[9] = {c00 * id, -(b * i9 - c * h) * id, (b * f - c * e) * id, c01 * id, (a * i9 - c * g) * id,
                         -(a * f - c * d) * id, c02 * id, -(a * h - b * g) * id, (a * e - b * d) * id};
  const double s01 = 0.5 * (inv[1] + inv[3]), s02 = 0.5 * (inv[2] + inv[6]), s12 = 0.5 * (inv[5] + inv[7]);
  double l00 = inv[0];
  if (!(l00 > 0.0)) return false;
  l00 = sqrt(l00);
  const double l10 = s01 / l00, l20 = s02 / l00;
  double l11 = inv[4] - l10 * l10;
  if (!(l11 > 0.0)) return false;
  l11 = sqrt(l11);
  const double l21 = (s12 - l20 * l10) / l11;
  double l22 = inv[8] - l20 * l20 - l21 * l21;
  if (!(l22 > 0.0)) return false;
  l22 = sqrt(l22);
  R[0] = l00; R[1] = l10; R[2] = l20; R[4] = l11; R[5] = l21; R[8] = l22;  // R = L^T
  return true;
}

// 0.5 * sum ||whitened error||^2 at values x (NonlinearFactorGraph::error)
__device__ __forceinline__ double s2_graph_error(const LmCommon &L, const double (*x)[3], int K, int nr, int lane) {
  double e = 0.0;
  if (lane < K) {
    const double r0 = x[lane][0] - L.mm[lane][0], r1 = x[lane][1] - L.mm[lane][1], r2 = x[lane][2] - L.mm[lane][2];
    for (int i = 0; i < 3; ++i) {
      const double w = L.Rm[lane][i * 3 + 0] * r0 + L.Rm[lane][i * 3 + 1] * r1 + L.Rm[lane][i * 3 + 2] * r2;
      e += 0.5 * w * w;
    }
  } else if (lane < K + nr) {
    const int ee = lane - K;
    const int v1 = L.rv1[ee], v2 = L.rv2[ee];
    const double dx = x[v2][0] - x[v1][0], dy = x[v2][1] - x[v1][1], dz = x[v2][2] - x[v1][2];
    const double r = sqrt(dx * dx + dy * dy + dz * dz);
    const double w = (r - L.rL[ee]) / L.rS[ee];
    e = 0.5 * w * w;
  }
  return s2_wave_sum(e);
}

// linearise at L.xc: b = -whitened error; range rows keep H/sigma (RangeFactor<Point3>: d/r, or (1,1,1) if r <= 1e-10)
__device__ __forceinline__ void s2_linearize(LmCommon &L, int K, int nr, int lane) {
  if (lane < K) {
    const double r0 = L.xc[lane][0] - L.mm[lane][0], r1 = L.xc[lane][1] - L.mm[lane][1], r2 = L.xc[lane][2] - L.mm[lane][2];
    for (int i = 0; i < 3; ++i)
      L.bb[3 * lane + i] = -(L.Rm[lane][i * 3 + 0] * r0 + L.Rm[lane][i * 3 + 1] * r1 + L.Rm[lane][i * 3 + 2] * r2);
  } else if (lane < K + nr) {
    const int ee = lane - K;
    const int v1 = L.rv1[ee], v2 = L.rv2[ee];
    const double dx = L.xc[v2][0] - L.xc[v1][0], dy = L.xc[v2][1] - L.xc[v1][1], dz = L.xc[v2][2] - L.xc[v1][2];
    const double r = sqrt(dx * dx + dy * dy + dz * dz);
    double h0 = 1.0, h1 = 1.0, h2 = 1.0;
    if (fabs(r) > 1e-10) { h0 = dx / r; h1 = dy / r; h2 = dz / r; }
    const double sg = L.rS[ee];
    L.Hs[ee][0] = h0 / sg; L.Hs[ee][1] = h1 / sg; L.Hs[ee][2] = h2 / sg;
    L.bb[3 * K + ee] = -(r - L.rL[ee]) / sg;
  }
  __builtin_amdgcn_wave_barrier();
}

// dense J^T J on the matrix cores: operands generated on the fly, result (lower+upper) stored to L.Hm, diag += lambda
__device__ void s2_jtj_mfma(LmLds &L, int K, int nr, double lambda, int lane) {
  const int n = 3 * K, mrows = n + nr;
  const int ksteps = (mrows + 3) >> 2;
  const int ntile = (n + 15) >> 4;
  int cv[4], ca[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int c = 16 * t + (lane & 15);
    cv[t] = (c < n) ? c / 3 : -1;
    ca[t] = c % 3;
  }
  d4 acc[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) acc[i] = (d4){0.0, 0.0, 0.0, 0.0};
  for (int ks = 0; ks < ksteps; ++ks) {
    const int r = 4 * ks + (lane >> 4);
    double val[4] = {0.0, 0.0, 0.0, 0.0};
    if (r < n) {
      const int u = r / 3, ii = r - 3 * u;
#pragma unroll
      for (int t = 0; t < 4; ++t) val[t] = (cv[t] == u) ? L.Rm[u][ii * 3 + ca[t]] : 0.0;
    } else if (r < mrows) {
      const int ee = r - n;
      const int v1 = L.rv1[ee], v2 = L.rv2[ee];
#pragma unroll
      for (int t = 0; t < 4; ++t) val[t] = (cv[t] == v1) ? -L.Hs[ee][ca[t]] : (cv[t] == v2) ? L.Hs[ee][ca[t]] : 0.0;
    }
    acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(val[0], val[0], acc[0], 0, 0, 0);
    if (ntile > 1) {
      acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(val[0], val[1], acc[1], 0, 0, 0);
      acc[4] = __builtin_amdgcn_mfma_f64_16x16x4f64(val[1], val[1], acc[4], 0, 0, 0);
    }
    if (ntile > 2) {
      acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(val[0], val[2], acc[2], 0, 0, 0);
      acc[5] = __builtin_amdgcn_mfma_f64_16x16x4f64(val[1], val[2], acc[5], 0, 0, 0);
      acc[7] = __builtin_amdgcn_mfma_f64_16x16x4f64(val[2], val[2], acc[7], 0, 0, 0);
    }
    if (ntile > 3) {
      acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(val[0], val[3], acc[3], 0, 0, 0);
      acc[6] = __builtin_amdgcn_mfma_f64_16x16x4f64(val[1], val[3], acc[6], 0, 0, 0);
      acc[8] = __builtin_amdgcn_mfma_f64_16x16x4f64(val[2], val[3], acc[8], 0, 0, 0);
      acc[9] = __builtin_amdgcn_mfma_f64_16x16x4f64(val[3], val[3], acc[9], 0, 0, 0);
    }
  }
  // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
  const int ti_of[10] = {0, 0, 0, 0, 1, 1, 1, 2, 2, 3}, tj_of[10] = {0, 1, 2, 3, 1, 2, 3, 2, 3, 3};
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const int ti = ti_of[i], tj = tj_of[i];
    if (tj < ntile) {
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int rr = 16 * ti + (lane >> 4) + 4 * reg, cc = 16 * tj + (lane & 15);
        double v = acc[i][reg];
        if (rr == cc && rr < n) v += lambda;  // sqrt(lambda) * I prior per variable (LM damping, diagonalDamping = false)
        L.Hm[rr * HS + cc] = v;
        if (ti != tj) L.Hm[cc * HS + rr] = v;
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// in-place lower Cholesky of L.Hm[0..n) (left-looking; lane i owns row i).  Returns false if not positive definite.
__device__ bool s2_cholesky(double *Hm, int n, int lane) {
  for (int j = 0; j < n; ++j) {
    double sreg = 0.0;
    if (lane >= j && lane < n) {
      sreg = Hm[lane * HS + j];
      for (int k = 0; k < j; ++k) sreg -= Hm[lane * HS + k] * Hm[j * HS + k];
    }
    const double d = __shfl(sreg, j);
    if (!(d > 0.0)) return false;
    const double sd = sqrt(d);
    if (lane == j) Hm[j * HS + j] = sd;
    else if (lane > j && lane < n) Hm[lane * HS + j] = sreg / sd;
    __builtin_amdgcn_wave_barrier();
  }
  return true;
}

// solve L L^T x = rhs; lane i holds rhs[i] on entry and x[i] on return (column-oriented substitutions in registers)
__device__ double s2_chol_solve(const double *Hm, int n, double rhs, int lane) {
  double b = rhs;
  for (int k = 0; k < n; ++k) {
    const double yk = __shfl(b / Hm[k * HS + k], k);
    if (lane == k) b = yk;
    else if (lane > k && lane < n) b -= Hm[lane * HS + k] * yk;
  }
  for (int k = n - 1; k >= 0; --k) {
    const double xk = __shfl(b / Hm[k * HS + k], k);
    if (lane == k) b = xk;
    else if (lane < k) b -= Hm[k * HS + lane] * xk;
  }
  return b;
}

// in-place inverse of the lower-triangular factor (X = L^-1), lane i owns row i
__device__ void s2_tri_inverse(double *Hm, int n, int lane) {
  for (int j = n - 1; j >= 0; --j) {
    const double xjj = 1.0 / Hm[j * HS + j];
    double tacc = 0.0;
    if (lane > j && lane < n)
      for (int k = j + 1; k <= lane; ++k) tacc += Hm[lane * HS + k] * Hm[k * HS + j];
    __builtin_amdgcn_wave_barrier();
    if (lane > j && lane < n) Hm[lane * HS + j] = -tacc * xjj;
    if (lane == j) Hm[j * HS + j] = xjj;
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Schur / Woodbury form of the damped normal equations.  H = D + W W^T with D = blockdiag(A_v + lambda I) (A_v = R_v^T R_v,
// the unary information of joint v) and W the whitened bone-length Jacobian (one column per bone factor, 6 non-zeros):
//   (D + W W^T)^-1 g = z - D^-1 W M^-1 W^T z,   z = D^-1 g,   M = I + W^T (D^-1 W)   (<= 21 x 21, SPD, well conditioned)
// M - I = W^T U (U = D^-1 W) is the J^T J of the bone factors in the unary metric and is built on the matrix cores
// (v_mfma_f64_16x16x4_f64, operands generated on the fly); its Cholesky factor, the triangular solves and the inverse
// live entirely in registers (lane i owns row i, L[i][k] broadcast with v_readlane).  Exact algebra: same solution as the
// dense 63 x 63 system up to rounding.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double s2_rl(double v, int src) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, src);
  hi = __builtin_amdgcn_readlane(hi, src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double s2_sel3(const double a[3], int i) { return i == 0 ? a[0] : (i == 1 ? a[1] : a[2]); }
__device__ __forceinline__ void s2_sym3_inv(const double a[6], double o[6]) {
  const double c00 = a[3] * a[5] - a[4] * a[4], c01 = -(a[1] * a[5] - a[4] * a[2]), c02 = a[1] * a[4] - a[3] * a[2];
  const double det = a[0] * c00 + a[1] * c01 + a[2] * c02, id = 1.0 / det;
  o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id;
  o[3] = (a[0] * a[5] - a[2] * a[2]) * id; o[4] = -(a[0] * a[4] - a[1] * a[2]) * id; o[5] = (a[0] * a[3] - a[1] * a[1]) * id;
}
__device__ __forceinline__ void s2_sym3_mul(const double m[6], const double v[3], double o[3]) {
  o[0] = m[0] * v[0] + m[1] * v[1] + m[2] * v[2];
  o[1] = m[1] * v[0] + m[3] * v[1] + m[4] * v[2];
  o[2] = m[2] * v[0] + m[4] * v[1] + m[5] * v[2];
}

// D^-1, z = D^-1 g, U = D^-1 W, M = I + W^T U (MFMA) -> row[] = row `lane` of M (identity rows beyond nr)
template <class LT>
__device__ void s2_schur_build(LT &L, int K, int nr, double lambda, int lane, const double Av[6], const double gv[3], double Di[6],
                               double zv[3], double row[21]) {
  if (lane < K) {
    const double D[6] = {Av[0] + lambda, Av[1], Av[2], Av[3] + lambda, Av[4], Av[5] + lambda};
    s2_sym3_inv(D, Di);
    s2_sym3_mul(Di, gv, zv);
    for (int q = 0; q < 6; ++q) L.Dv[lane][q] = Di[q];
    for (int a = 0; a < 3; ++a) L.zz[lane][a] = zv[a];
  }
  __builtin_amdgcn_wave_barrier();
  if (lane < nr) {
    const int v1 = L.rv1[lane], v2 = L.rv2[lane];
    const double hs[3] = {L.Hs[lane][0], L.Hs[lane][1], L.Hs[lane][2]};
    double d1[6], d2[6], o1[3], o2[3];
    for (int q = 0; q < 6; ++q) { d1[q] = L.Dv[v1][q]; d2[q] = L.Dv[v2][q]; }
    s2_sym3_mul(d1, hs, o1);
    s2_sym3_mul(d2, hs, o2);
    for (int a = 0; a < 3; ++a) { L.ua[lane][a] = -o1[a]; L.ub[lane][a] = o2[a]; }
  }
  __builtin_amdgcn_wave_barrier();
  const int nt = (nr + 15) >> 4, ksteps = (3 * K + 3) >> 2;
  int ev1[2], ev2[2];
  double hs_[2][3], ua_[2][3], ub_[2][3];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int e = 16 * t + (lane & 15);
    const bool ok = e < nr;
    ev1[t] = ok ? L.rv1[e] : -1;
    ev2[t] = ok ? L.rv2[e] : -1;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      hs_[t][a] = ok ? L.Hs[e][a] : 0.0;
      ua_[t][a] = ok ? L.ua[e][a] : 0.0;
      ub_[t][a] = ok ? L.ub[e][a] : 0.0;
    }
  }
  d4 a00 = {0, 0, 0, 0}, a01 = {0, 0, 0, 0}, a10 = {0, 0, 0, 0}, a11 = {0, 0, 0, 0};
  for (int ks = 0; ks < ksteps; ++ks) {
    const int r = 4 * ks + (lane >> 4);
    const int v = (r < 3 * K) ? r / 3 : -2;
    const int ii = r - 3 * (r / 3);
    double w[2], u[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const double h = s2_sel3(hs_[t], ii);
      w[t] = (v == ev1[t]) ? -h : (v == ev2[t]) ? h : 0.0;
      u[t] = (v == ev1[t]) ? s2_sel3(ua_[t], ii) : (v == ev2[t]) ? s2_sel3(ub_[t], ii) : 0.0;
    }
    a00 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[0], u[0], a00, 0, 0, 0);
    if (nt > 1) {
      a01 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[0], u[1], a01, 0, 0, 0);
      a10 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[1], u[0], a10, 0, 0, 0);
      a11 = __builtin_amdgcn_mfma_f64_16x16x4f64(w[1], u[1], a11, 0, 0, 0);
    }
  }
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) {  // C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
    const int rr = (lane >> 4) + 4 * reg, cc = lane & 15;
    L.Mm[rr][cc] = a00[reg] + (rr == cc ? 1.0 : 0.0);
    L.Mm[rr][cc + 16] = a01[reg];
    L.Mm[rr + 16][cc] = a10[reg];
    L.Mm[rr + 16][cc + 16] = a11[reg] + (rr == cc ? 1.0 : 0.0);
  }
  __builtin_amdgcn_wave_barrier();
  const int li = lane < 32 ? lane : 31;
#pragma unroll
  for (int k = 0; k < 21; ++k) row[k] = (lane < nr) ? L.Mm[li][k] : (k == lane ? 1.0 : 0.0);
}

// in-register Cholesky of the (<= 21 x 21) matrix whose row `lane` is row[]; on return row[k] = L[lane][k] for k <= lane
__device__ __forceinline__ void s2_chol21(double row[21], int lane) {
#pragma unroll
  for (int j = 0; j < 21; ++j) {
    double sreg = row[j];
#pragma unroll
    for (int k = 0; k < j; ++k) sreg -= row[k] * s2_rl(row[k], j);
    const double sd = sqrt(s2_rl(sreg, j));
    row[j] = (lane == j) ? sd : ((lane > j) ? sreg / sd : row[j]);
  }
}

// tryLambda solve: delta = (D + W W^T)^-1 g  -> L.dl
template <class LT>
__device__ void s2_schur_solve(LT &L, int K, int nr, double lambda, int lane, const double Av[6], const double gv[3]) {
  double Di[6], zv[3] = {0, 0, 0}, row[21];
  s2_schur_build(L, K, nr, lambda, lane, Av, gv, Di, zv, row);
  double b = 0.0;
  if (lane < nr) {
    const int v1 = L.rv1[lane], v2 = L.rv2[lane];
    b = L.Hs[lane][0] * (L.zz[v2][0] - L.zz[v1][0]) + L.Hs[lane][1] * (L.zz[v2][1] - L.zz[v1][1]) + L.Hs[lane][2] * (L.zz[v2][2] - L.zz[v1][2]);
  }
  s2_chol21(row, lane);
#pragma unroll
  for (int k = 0; k < 21; ++k) {  // forward substitution, column oriented
    const double yk = s2_rl(b, k) / s2_rl(row[k], k);
    b = (lane == k) ? yk : ((lane > k) ? b - row[k] * yk : b);
  }
  if (lane < 21) {
#pragma unroll
    for (int k = 0; k < 21; ++k) L.Mm[lane][k] = row[k];
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int k = 20; k >= 0; --k) {  // backward substitution with L^T (factor staged in LDS)
    const double xk = s2_rl(b, k) / s2_rl(row[k], k);
    const double lkl = (lane < k) ? L.Mm[k][lane] : 0.0;
    b = (lane == k) ? xk : ((lane < k) ? b - lkl * xk : b);
  }
  if (lane < 32) L.yy[lane] = b;
  __builtin_amdgcn_wave_barrier();
  if (lane < K) {
    double d[3] = {zv[0], zv[1], zv[2]};
    for (unsigned m = L.incA[lane]; m; m &= m - 1) {
      const int e = __ffs(m) - 1;
      const double y = L.yy[e];
      for (int a = 0; a < 3; ++a) d[a] -= L.ua[e][a] * y;
    }
    for (unsigned m = L.incB[lane]; m; m &= m - 1) {
      const int e = __ffs(m) - 1;
      const double y = L.yy[e];
      for (int a = 0; a < 3; ++a) d[a] -= L.ub[e][a] * y;
    }
    for (int a = 0; a < 3; ++a) L.dl[3 * lane + a] = d[a];
  }
  __builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------------------------------------------------------
// Forest elimination.  The active bone factors always form a forest (the skeleton tree of addBinaryFactors, PP:384-481:
// MidHip-Neck is only active when Belly is absent, so no cycle can appear), hence the damped normal equations
//   H: diagonal blocks A_v + lambda I + sum_{e at v} hs_e hs_e^T, off-diagonal blocks -hs_e hs_e^T on the edges
// have a perfect (fill-free) elimination order: leaves first.  Eliminating child c into parent p over edge h = hs_e:
//   D_p -= (h^T D_c^-1 h) h h^T,   g_p += (h^T D_c^-1 g_c) h;   back-substitution  d_c = D_c^-1 g_c + D_c^-1 h (h^T d_p).
// This is the exact block Cholesky solve a multifrontal solver performs with a perfect ordering; lane v owns joint v, all
// joints of one depth are processed together and only 3x3 symmetric inverses are needed.
// c_tree_rank[k] = position of fusion joint k in a breadth-first order of the union skeleton tree rooted at the belly
// (20,1,8,0,2,5,9,12,3,6,10,13,15,16,19,4,7,11,14,17,18): every joint's possible parents precede it.  Rooted there the forest is
// four levels deep (belly-neck-shoulder-elbow-wrist, belly-midhip-hip-knee-ankle; without the belly: neck-midhip-hip-knee-ankle),
// one less than from the MidHip: the levels are the sequential part of every solve.
// ---------------------------------------------------------------------------------------------------------------
__device__ __constant__ const int c_tree_rank[SE3D_NFU] = {3, 1, 4, 8, 15, 5, 9, 16, 2, 6, 10, 17, 7, 11, 18, 12, 13, 19, 20, 14, 0};

struct TreeTopo { int parent, pedge, level, maxlev; unsigned children; bool forest; };

__device__ TreeTopo s2_tree_topology(LmSchurBase &L, int K, int nr, int lane) {
  TreeTopo tp{-1, -1, 0, 0, 0u, true};
  int nparents = 0;
  if (lane < K) {
    const int myrank = c_tree_rank[L.keyofcol[lane]];
    for (unsigned m = L.incA[lane] | L.incB[lane]; m; m &= m - 1) {
      const int e = __ffs(m) - 1;
      const int other = (L.rv1[e] == lane) ? L.rv2[e] : L.rv1[e];
      if (c_tree_rank[L.keyofcol[other]] < myrank) { tp.parent = other; tp.pedge = e; ++nparents; }
      else tp.children |= 1u << other;
    }
    L.par[lane] = tp.parent;
    L.pe[lane] = tp.pedge;
  }
  tp.forest = __ballot(nparents > 1) == 0ull;  // cannot fail with the reference's bone tables; guards the assumption
  __builtin_amdgcn_wave_barrier();
  if (lane < K) {
    int lv = 0, p = tp.parent;
    for (int hop = 0; hop < SE3D_NFU && p >= 0; ++hop) { ++lv; p = L.par[p]; }
    tp.level = lv;
  }
  int mx = (lane < K) ? tp.level : 0;
  for (int o = 32; o > 0; o >>= 1) { const int t = __shfl_xor(mx, o); mx = t > mx ? t : mx; }
  tp.maxlev = mx;
  return tp;
}

// tryLambda solve by forest elimination: delta -> L.dl
__device__ void s2_tree_solve(LmSchurBase &L, int K, double lambda, int lane, const double Av[6], const double gv[3], const TreeTopo &tp) {
  double D[6] = {Av[0] + lambda, Av[1], Av[2], Av[3] + lambda, Av[4], Av[5] + lambda};
  double g[3] = {gv[0], gv[1], gv[2]};
  double Di[6] = {0, 0, 0, 0, 0, 0}, u[3] = {0, 0, 0}, z[3] = {0, 0, 0}, h[3] = {0, 0, 0};
  const bool act = lane < K;
  if (act) {
    for (unsigned m = L.incA[lane] | L.incB[lane]; m; m &= m - 1) {
      const int e = __ffs(m) - 1;
      const double h0 = L.Hs[e][0], h1 = L.Hs[e][1], h2 = L.Hs[e][2];
      D[0] += h0 * h0; D[1] += h0 * h1; D[2] += h0 * h2; D[3] += h1 * h1; D[4] += h1 * h2; D[5] += h2 * h2;
    }
    if (tp.pedge >= 0) { h[0] = L.Hs[tp.pedge][0]; h[1] = L.Hs[tp.pedge][1]; h[2] = L.Hs[tp.pedge][2]; }
  }
  for (int lv = tp.maxlev; lv >= 1; --lv) {  // leaves -> roots
    if (act && tp.level == lv) {
      s2_sym3_inv(D, Di);
      s2_sym3_mul(Di, h, u);
      s2_sym3_mul(Di, g, z);
      L.al[lane] = h[0] * u[0] + h[1] * u[1] + h[2] * u[2];
      L.be[lane] = h[0] * z[0] + h[1] * z[1] + h[2] * z[2];
    }
    __builtin_amdgcn_wave_barrier();
    if (act && tp.level == lv - 1) {
      for (unsigned m = tp.children; m; m &= m - 1) {
        const int c = __ffs(m) - 1;
        const int e = L.pe[c];
        const double h0 = L.Hs[e][0], h1 = L.Hs[e][1], h2 = L.Hs[e][2];
        const double a = L.al[c], b = L.be[c];
        D[0] -= a * h0 * h0; D[1] -= a * h0 * h1; D[2] -= a * h0 * h2; D[3] -= a * h1 * h1; D[4] -= a * h1 * h2; D[5] -= a * h2 * h2;
        g[0] += b * h0; g[1] += b * h1; g[2] += b * h2;
      }
    }
  }
  double d[3] = {0, 0, 0};
  if (act && tp.level == 0) {
    s2_sym3_inv(D, Di);
    s2_sym3_mul(Di, g, d);
    for (int a = 0; a < 3; ++a) L.dl[3 * lane + a] = d[a];
  }
  for (int lv = 1; lv <= tp.maxlev; ++lv) {  // roots -> leaves
    __builtin_amdgcn_wave_barrier();
    if (act && tp.level == lv) {
      const int p = tp.parent;
      const double gam = h[0] * L.dl[3 * p] + h[1] * L.dl[3 * p + 1] + h[2] * L.dl[3 * p + 2];
      for (int a = 0; a < 3; ++a) { d[a] = z[a] + u[a] * gam; L.dl[3 * lane + a] = d[a]; }
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// Marginal covariances on the forest (lambda = 0): after the leaf-to-root elimination D~_c is the information of joint c
// given its parent, so  Sigma_root = D~_root^-1  and  Sigma_cc = D~_c^-1 + u (h^T Sigma_pp h) u^T,  u = D~_c^-1 h  (top-down).
// Same 3x3 diagonal blocks of (J^T J)^-1 as Marginals::marginalCovariance (PP:789), exact algebra.
__device__ void s2_tree_marginals(LmSchurBase &L, int K, int lane, const double Av[6], const TreeTopo &tp) {
  double D[6] = {Av[0], Av[1], Av[2], Av[3], Av[4], Av[5]};
  double Di[6] = {0, 0, 0, 0, 0, 0}, u[3] = {0, 0, 0}, h[3] = {0, 0, 0};
  const bool act = lane < K;
  if (act) {
    for (unsigned m = L.incA[lane] | L.incB[lane]; m; m &= m - 1) {
      const int e = __ffs(m) - 1;
      const double h0 = L.Hs[e][0], h1 = L.Hs[e][1], h2 = L.Hs[e][2];
      D[0] += h0 * h0; D[1] += h0 * h1; D[2] += h0 * h2; D[3] += h1 * h1; D[4] += h1 * h2; D[5] += h2 * h2;
    }
    if (tp.pedge >= 0) { h[0] = L.Hs[tp.pedge][0]; h[1] = L.Hs[tp.pedge][1]; h[2] = L.Hs[tp.pedge][2]; }
  }
  for (int lv = tp.maxlev; lv >= 1; --lv) {
    if (act && tp.level == lv) {
      s2_sym3_inv(D, Di);
      s2_sym3_mul(Di, h, u);
      L.al[lane] = h[0] * u[0] + h[1] * u[1] + h[2] * u[2];
    }
    __builtin_amdgcn_wave_barrier();
    if (act && tp.level == lv - 1) {
      for (unsigned m = tp.children; m; m &= m - 1) {
        const int c = __ffs(m) - 1;
        const int e = L.pe[c];
        const double h0 = L.Hs[e][0], h1 = L.Hs[e][1], h2 = L.Hs[e][2];
        const double a = L.al[c];
        D[0] -= a * h0 * h0; D[1] -= a * h0 * h1; D[2] -= a * h0 * h2; D[3] -= a * h1 * h1; D[4] -= a * h1 * h2; D[5] -= a * h2 * h2;
      }
    }
  }
  if (act && tp.level == 0) {
    s2_sym3_inv(D, Di);
    for (int q = 0; q < 6; ++q) L.mcov[lane][q] = Di[q];
  }
  for (int lv = 1; lv <= tp.maxlev; ++lv) {
    __builtin_amdgcn_wave_barrier();
    if (act && tp.level == lv) {
      double sp[6], sh[3];
      for (int q = 0; q < 6; ++q) sp[q] = L.mcov[tp.parent][q];
      s2_sym3_mul(sp, h, sh);
      const double gam = h[0] * sh[0] + h[1] * sh[1] + h[2] * sh[2];
      L.mcov[lane][0] = Di[0] + gam * u[0] * u[0]; L.mcov[lane][1] = Di[1] + gam * u[0] * u[1]; L.mcov[lane][2] = Di[2] + gam * u[0] * u[2];
      L.mcov[lane][3] = Di[3] + gam * u[1] * u[1]; L.mcov[lane][4] = Di[4] + gam * u[1] * u[2]; L.mcov[lane][5] = Di[5] + gam * u[2] * u[2];
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// Marginals: (H^-1)_vv = D_v^-1 - sum_{e,f incident to v} u_e (M^-1)[e][f] u_f^T  -> L.mcov
template <class LT>
__device__ void s2_schur_marginals(LT &L, int K, int nr, int lane, const double Av[6]) {
  double Di[6] = {0, 0, 0, 0, 0, 0}, zv[3], row[21];
  const double g0[3] = {0, 0, 0};
  s2_schur_build(L, K, nr, 0.0, lane, Av, g0, Di, zv, row);
  s2_chol21(row, lane);
  double x[21];  // column `lane` of X = L^-1
#pragma unroll
  for (int i = 0; i < 21; ++i) {
    double sacc = (i == lane) ? 1.0 : 0.0;
#pragma unroll
    for (int k = 0; k < i; ++k) sacc -= s2_rl(row[k], i) * x[k];
    x[i] = sacc / s2_rl(row[i], i);
  }
#pragma unroll
  for (int f = 0; f < 21; ++f) {  // (M^-1)[lane][f] = sum_{i >= f} X[i][lane] X[i][f]
    double acc = 0.0;
#pragma unroll
    for (int i = f; i < 21; ++i) acc += x[i] * s2_rl(x[i], f);
    if (lane < 21) L.Mi[lane][f] = acc;
  }
  __builtin_amdgcn_wave_barrier();
  if (lane < K) {
    double c[6] = {Di[0], Di[1], Di[2], Di[3], Di[4], Di[5]};
    const unsigned mA = L.incA[lane], mB = L.incB[lane];
    for (unsigned me = mA | mB; me; me &= me - 1) {
      const int e = __ffs(me) - 1;
      const double *ue = ((mA >> e) & 1u) ? L.ua[e] : L.ub[e];
      for (unsigned mf = mA | mB; mf; mf &= mf - 1) {
        const int f = __ffs(mf) - 1;
        const double *uf = ((mA >> f) & 1u) ? L.ua[f] : L.ub[f];
        const double mi = L.Mi[e][f];
        c[0] -= mi * ue[0] * uf[0]; c[1] -= mi * ue[0] * uf[1]; c[2] -= mi * ue[0] * uf[2];
        c[3] -= mi * ue[1] * uf[1]; c[4] -= mi * ue[1] * uf[2]; c[5] -= mi * ue[2] * uf[2];
      }
    }
    for (int q = 0; q < 6; ++q) L.mcov[lane][q] = c[q];
  }
  __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void s2_cov6_to_mat(const double c[6], double M[9]) {
  M[0] = c[0]; M[1] = c[1]; M[2] = c[2]; M[3] = c[1]; M[4] = c[3]; M[5] = c[4]; M[6] = c[2]; M[7] = c[4]; M[8] = c[5];
}

// MODE 0: dense 63x63 H per wavefront (J^T J on MFMA); (1: Schur / Woodbury on MFMA for solve and marginals — removed in round 5, 0.92 ms against 0.30 ms; its solve stays as the non-forest fallback of modes 2 / 3); 2: forest elimination
// for the LM solves + Schur/Woodbury (MFMA) marginals; 3: forest elimination for solves and marginals (no matrix product left)
template <int MODE, class LT = typename std::conditional<MODE != 0, LmSchurLds, LmLds>::type>
__device__ __forceinline__ void s2_lm_body(const Se3dStage2 &Q, int i, int s, int lane, LT &L) {  // one wavefront
  constexpr bool SCHUR = MODE != 0;
  const int H = Q.H, T = Q.T;
  if (i >= min(Q.n_det[s], H)) return;
  const int slot = Q.det_slot[(size_t)s * H + i];
  if (slot < 0) return;
  const size_t sl = (size_t)s * T + slot;
  const se3d_person3d &person = Q.dets[(size_t)s * H + i];
  const bool simple = Q.pose_method == SE3D_POSE_SIMPLE;
  const double t = Q.t[s];
  LM_STAMP_INIT

  // ---- root / neck / height (PP:631-668), identical on every lane ----
  double root[3] = {0, 0, 0}, neck[3] = {0, 0, 0};
  float root_score = 0.f, neck_score = 0.f;
  if (!simple) {
    for (int a = 0; a < 3; ++a) { root[a] = person.keypoints[FU_MidHip].joint[a]; neck[a] = person.keypoints[FU_Neck].joint[a]; }
    root_score = person.keypoints[FU_MidHip].score;
    neck_score = person.keypoints[FU_Neck].score;
  } else {
    const se3d_keypoint3d &hl = person.keypoints[FU_LHip], &hr = person.keypoints[FU_RHip];
    if (hl.score > 0.0f && hr.score > 0.0f) {
      for (int a = 0; a < 3; ++a) root[a] = (hl.joint[a] + hr.joint[a]) / 2.0;
      root_score = (hl.score + hr.score) / 2.0f;
    }
    const se3d_keypoint3d &sL = person.keypoints[FU_LShoulder], &sR = person.keypoints[FU_RShoulder];
    if (sL.score > 0.0f && sR.score > 0.0f) {
      for (int a = 0; a < 3; ++a) neck[a] = (sL.joint[a] + sR.joint[a]) / 2.0;
      neck_score = (sL.score + sR.score) / 2.0f;
    }
  }
  const bool has_root = root_score > 0.10f;
  double height = 1.0;
  if (has_root && Q.norm_height) {
    if (neck_score > 0.10f) {
      const double dx = neck[0] - root[0], dy = neck[1] - root[1], dz = neck[2] - root[2];
      height = sqrt(dx * dx + dy * dy + dz * dz);
    } else height = 0.60;
  }
  if (Q.height_prev[sl] < 0.0) {  // PP:699-702 (uniform; all lanes write the same values)
    if (lane == 0) { Q.height_prev[sl] = height; for (int a = 0; a < 3; ++a) Q.root_prev[sl * 3 + a] = root[a]; }
  }
  __builtin_amdgcn_wave_barrier();
  const double hprev = (Q.height_prev[sl] < 0.0) ? height : Q.height_prev[sl];
  double rprev[3];
  for (int a = 0; a < 3; ++a) rprev[a] = (Q.height_prev[sl] < 0.0) ? root[a] : Q.root_prev[sl * 3 + a];

  // ---- measurements: lane k <-> fusion joint k (PP:690-737) ----
  bool measured = false;
  double m[3] = {0, 0, 0}, cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  float kscore = 0.f;
  if (lane < SE3D_NFU) {
    const int k = lane;
    if (k == FU_MidHip) {
      if (has_root) {
        measured = true;
        if (!simple) s2_cov6_to_mat(person.keypoints[FU_MidHip].cov, cov);
        else {
          double cl[9], cr[9];
          s2_cov6_to_mat(person.keypoints[FU_LHip].cov, cl);
          s2_cov6_to_mat(person.keypoints[FU_RHip].cov, cr);
          for (int q = 0; q < 9; ++q) cov[q] = (cl[q] + cr[q]) / 2.0;
        }
        for (int q = 0; q < 9; ++q) cov[q] = cov[q] / (height * height) / (100.0 * 100.0);
        kscore = root_score;
      }
    } else if (simple && k == FU_Neck && neck_score > 0.10f) {
      measured = true;
      double cl[9], cr[9];
      s2_cov6_to_mat(person.keypoints[FU_LShoulder].cov, cl);
      s2_cov6_to_mat(person.keypoints[FU_RShoulder].cov, cr);
      for (int q = 0; q < 9; ++q) cov[q] = (cl[q] + cr[q]) / 2.0 / (height * height);
      for (int a = 0; a < 3; ++a) m[a] = (neck[a] - root[a]) / height;
      kscore = neck_score;
    } else {
      const se3d_keypoint3d &kp = person.keypoints[k];
      if (kp.score > 0.10f) {
        measured = true;
        double c9[9];
        s2_cov6_to_mat(kp.cov, c9);
        for (int q = 0; q < 9; ++q) cov[q] = c9[q] / (height * height);
        for (int a = 0; a < 3; ++a) m[a] = (kp.joint[a] - root[a]) / height;
        kscore = kp.score;
      }
    }
  }
  const unsigned long long mb = __ballot(measured);
  const uint32_t measmask = (uint32_t)mb;
  const int K = __popcll(mb);
  if (K == 0) return;  // PP:739-741
  const int col = __popcll(mb & ((1ull << lane) - 1ull));
  const uint32_t hasprev = Q.has[sl];
  const bool use_vel = measured && ((hasprev >> lane) & 1u);
  bool wok = true;
  double x0[3] = {0, 0, 0};
  if (measured) {
    double R[9];
    wok = s2_make_whitener(cov, R);
    for (int q = 0; q < 9; ++q) L.Rm[col][q] = R[q];
    for (int a = 0; a < 3; ++a) {
      L.mm[col][a] = m[a];
      x0[a] = use_vel ? Q.prev[(sl * SE3D_NFU + lane) * 3 + a] : m[a];  // setInitialState (PP:483-503)
      L.xc[col][a] = x0[a];
    }
    L.keyofcol[col] = lane;
  }
  const bool whiten_ok = __ballot(!wok) == 0ull;
  // ---- range factors in table order (PP:384-481) ----
  bool ract = false;
  Se3dBone bd = c_bones[Q.norm_height ? 1 : 0][lane < 21 ? lane : 0];
  if (lane < 21) ract = ((measmask >> bd.k1) & 1u) && ((measmask >> bd.k2) & 1u) && (!bd.not_belly || !((measmask >> FU_Belly) & 1u));
  const unsigned long long rb = __ballot(ract);
  const int nr = __popcll(rb);
  if (ract) {
    const int ee = __popcll(rb & ((1ull << lane) - 1ull));
    L.rv1[ee] = __popc(measmask & ((1u << bd.k1) - 1u));
    L.rv2[ee] = __popc(measmask & ((1u << bd.k2) - 1u));
    L.rL[ee] = bd.L;
    L.rS[ee] = bd.sigma * (Q.norm_height ? 2.0 : 1.0);  // g_limbLSigmaFactor (PP:934-937)
  }
  __builtin_amdgcn_wave_barrier();
  const int n = 3 * K;
  double Av[6] = {0, 0, 0, 0, 0, 0};  // Schur variant: lane v < K owns variable (column) v
  if constexpr (SCHUR) {
    if (lane < K) {
      const double *R = L.Rm[lane];
      Av[0] = R[0] * R[0] + R[3] * R[3] + R[6] * R[6]; Av[1] = R[0] * R[1] + R[3] * R[4] + R[6] * R[7];
      Av[2] = R[0] * R[2] + R[3] * R[5] + R[6] * R[8]; Av[3] = R[1] * R[1] + R[4] * R[4] + R[7] * R[7];
      Av[4] = R[1] * R[2] + R[4] * R[5] + R[7] * R[8]; Av[5] = R[2] * R[2] + R[5] * R[5] + R[8] * R[8];
      unsigned ia = 0, ib = 0;
      for (int ee = 0; ee < nr; ++ee) { if (L.rv1[ee] == lane) ia |= 1u << ee; if (L.rv2[ee] == lane) ib |= 1u << ee; }
      L.incA[lane] = ia; L.incB[lane] = ib;
    }
    __builtin_amdgcn_wave_barrier();
  }
  TreeTopo topo{-1, -1, 0, 0, 0u, false};
  if constexpr (MODE >= 2) topo = s2_tree_topology(L, K, nr, lane);

  LM_STAMP(0)  // setup
  // ---- LevenbergMarquardtOptimizer(graph, prevEstimate).optimize(), GTSAM 4.0.3 defaults ----
  int iterations = 0;
  double obj_final = -1.0;   // graph.error(result) of the published values (se3d_fused_out.lm_objective); -1: the optimisation failed
  if (whiten_ok) {
    double error = s2_graph_error(L, L.xc, K, nr, lane);
    double lambda = 1e-5;
    if (!(error <= 0.0) && !LM_ABLATE(4)) {
      for (int outer = 0; outer < 1000; ++outer) {
        const double currentError = error;
        LM_STAMP(1)  // graph error / loop control
        if (!LM_ABLATE(16)) s2_linearize(L, K, nr, lane);
        // g = J^T b (dense: lane c < n holds g[c]; Schur: lane v < K holds the 3-vector of its variable)
        double g = 0.0;
        double gv[3] = {0, 0, 0};
        if constexpr (SCHUR) {
          if (lane < K) {
            const double *R = L.Rm[lane];
            const double b0 = L.bb[3 * lane], b1 = L.bb[3 * lane + 1], b2 = L.bb[3 * lane + 2];
            for (int a = 0; a < 3; ++a) gv[a] = R[0 * 3 + a] * b0 + R[1 * 3 + a] * b1 + R[2 * 3 + a] * b2;
            for (unsigned mk = L.incA[lane]; mk; mk &= mk - 1) {
              const int ee = __ffs(mk) - 1;
              for (int a = 0; a < 3; ++a) gv[a] -= L.Hs[ee][a] * L.bb[n + ee];
            }
            for (unsigned mk = L.incB[lane]; mk; mk &= mk - 1) {
              const int ee = __ffs(mk) - 1;
              for (int a = 0; a < 3; ++a) gv[a] += L.Hs[ee][a] * L.bb[n + ee];
            }
          }
        } else if (lane < n) {
          const int v = lane / 3, a = lane - 3 * v;
          g = L.Rm[v][0 * 3 + a] * L.bb[3 * v] + L.Rm[v][1 * 3 + a] * L.bb[3 * v + 1] + L.Rm[v][2 * 3 + a] * L.bb[3 * v + 2];
          for (int ee = 0; ee < nr; ++ee) {
            const double hv = (v == L.rv1[ee]) ? -L.Hs[ee][a] : (v == L.rv2[ee]) ? L.Hs[ee][a] : 0.0;
            g += hv * L.bb[n + ee];
          }
        }
        bool stepped = false;
        for (int tr = 0; tr < 64; ++tr) {  // tryLambda
          bool solved = true;
          LM_STAMP(2)  // linearize + gradient
          if constexpr (MODE >= 2) {
            if (LM_ABLATE(8)) { /* diagnostic: keep the previous delta */ }
            else if (topo.forest) s2_tree_solve(L, K, lambda, lane, Av, gv, topo);
            else s2_schur_solve(L, K, nr, lambda, lane, Av, gv);
            LM_STAMP(3)  // solve
          } else {
            s2_jtj_mfma(L, K, nr, lambda, lane);
            solved = s2_cholesky(L.Hm, n, lane);
          }
          bool step_ok = false, stop = false;
          double newError = 0.0;
          if (solved) {
            if constexpr (!SCHUR) {
              const double d = s2_chol_solve(L.Hm, n, g, lane);
              if (lane < n) L.dl[lane] = d;
              __builtin_amdgcn_wave_barrier();
            }
            // linear.error(delta) = 0.5 ||J delta - b||^2 over the undamped factors
            double le = 0.0;
            for (int r = lane; r < n + nr && !LM_ABLATE(32); r += 64) {
              double sv;
              if (r < n) {
                const int u = r / 3, ii = r - 3 * u;
                sv = L.Rm[u][ii * 3 + 0] * L.dl[3 * u] + L.Rm[u][ii * 3 + 1] * L.dl[3 * u + 1] + L.Rm[u][ii * 3 + 2] * L.dl[3 * u + 2] - L.bb[r];
              } else {
                const int ee = r - n;
                const int v1 = L.rv1[ee], v2 = L.rv2[ee];
                sv = L.Hs[ee][0] * (L.dl[3 * v2] - L.dl[3 * v1]) + L.Hs[ee][1] * (L.dl[3 * v2 + 1] - L.dl[3 * v1 + 1]) +
                     L.Hs[ee][2] * (L.dl[3 * v2 + 2] - L.dl[3 * v1 + 2]) - L.bb[r];
              }
              le += 0.5 * sv * sv;
            }
            le = s2_wave_sum(le);
            const double linearizedCostChange = error - le;
            if (linearizedCostChange >= 0) {
              if (lane < K)
                for (int a = 0; a < 3; ++a) L.xn[lane][a] = L.xc[lane][a] + L.dl[3 * lane + a];  // retract
              __builtin_amdgcn_wave_barrier();
              newError = LM_ABLATE(64) ? error * 0.5 : s2_graph_error(L, L.xn, K, nr, lane);
              const double costChange = error - newError;
              if (linearizedCostChange > 1e-20) {
                const double modelFidelity = costChange / linearizedCostChange;
                step_ok = modelFidelity > 1e-3;
              }
              if (fabs(costChange) < 1e-5 * error) stop = true;
            }
          }
          if (step_ok) {  // decreaseLambda (useFixedLambdaFactor)
            if (lane < K)
              for (int a = 0; a < 3; ++a) L.xc[lane][a] = L.xn[lane][a];
            __builtin_amdgcn_wave_barrier();
            error = newError;
            lambda /= 10.0;
            if (lambda < 0.0) lambda = 0.0;
            ++iterations;
            stepped = true;
            break;
          } else if (!stop) {  // increaseLambda
            lambda *= 10.0;
            if (lambda >= 1e5) break;
          } else {
            break;
          }
        }
        (void)stepped;
        if (iterations >= 100) break;  // maxIterations
        if (LM_ABLATE(1)) break;
        if (error <= 0.0) break;       // errorTol
        const double absoluteDecrease = currentError - error;
        const double relativeDecrease = absoluteDecrease / currentError;
        if (relativeDecrease <= 1e-5 || absoluteDecrease <= 1e-5) break;
        if (!isfinite(currentError)) break;
      }
    }
    obj_final = error;
  } else {
    if (lane < K)
      for (int a = 0; a < 3; ++a) L.xc[lane][a] = L.mm[lane][a];  // optimisation failed: output the measurements (PP:751-758)
    __builtin_amdgcn_wave_barrier();
  }

  LM_STAMP(4)  // linear error, new error, lambda logic (accumulated remainder of the LM loop)
  // ---- Marginals(graph, result): diagonal 3x3 blocks of (J^T J)^-1 at the optimum (PP:760-793) ----
  bool use_marg = false;
  double mc[6] = {0, 0, 0, 0, 0, 0};
  if constexpr (SCHUR) {
    if (whiten_ok && !LM_ABLATE(2)) {
      s2_linearize(L, K, nr, lane);
      if (MODE == 3 && topo.forest) s2_tree_marginals(L, K, lane, Av, topo);
      else s2_schur_marginals(L, K, nr, lane, Av);
      use_marg = true;
      if (measured)
        for (int q = 0; q < 6; ++q) mc[q] = L.mcov[col][q];
    }
  } else if (whiten_ok) {
    s2_linearize(L, K, nr, lane);
    s2_jtj_mfma(L, K, nr, 0.0, lane);
    if (s2_cholesky(L.Hm, n, lane)) {
      use_marg = true;
      s2_tri_inverse(L.Hm, n, lane);
      // lane k (measured) gathers its own block: (H^-1)[3c+a][3c+b] = sum_{i >= 3c+b} X[i][3c+a] X[i][3c+b]
      if (measured) {
        const int c0 = 3 * col;
        int q = 0;
        for (int a = 0; a < 3; ++a)
          for (int b = a; b < 3; ++b, ++q) {
            double acc = 0.0;
            for (int r = c0 + b; r < n; ++r) acc += L.Hm[r * HS + c0 + a] * L.Hm[r * HS + c0 + b];
            mc[q] = acc;
          }
      }
    }
  }

  LM_STAMP(5)  // marginals
  // ---- outputs + track-state update: lane k <-> fusion joint k (PP:769-843) ----
  // direct_out (fused walk): a detection that will be published carries pub = 1 + its output slot since the association phase
  const int pubv = Q.direct_out ? (int)Q.pub[(size_t)s * H + i] : 0;
  se3d_person3d &pf = pubv ? Q.fused[(size_t)s * H + pubv - 1] : Q.fused_det[(size_t)s * H + i];
  se3d_person3d &pp = pubv ? Q.pred[(size_t)s * H + pubv - 1] : Q.pred_det[(size_t)s * H + i];
  {  // zero both persons cooperatively, then fill the measured joints
    uint32_t *w0 = reinterpret_cast<uint32_t *>(&pf), *w1 = reinterpret_cast<uint32_t *>(&pp);
    for (int w = lane; w < (int)(sizeof(se3d_person3d) / 4); w += 64) { w0[w] = 0u; w1[w] = 0u; }
  }
  __builtin_amdgcn_wave_barrier();
  LM_STAMP(7)  // (diag) outputs: zero fill
  const int track_id = Q.id[sl];
  if (lane == 0) { pf.id = (uint32_t)track_id; pp.id = (uint32_t)track_id; }
  const double pred_dt = Q.pred_dt[s];
  const int fslot = Q.frame_nr[s] % 3;
  const double gtprev = Q.t_prev[s];
#ifdef SE3D_LM_STAMPS
  if (pred_dt + gtprev + fslot + track_id == -12345.678) return;  // force the loads to complete before the stamp
#endif
  LM_STAMP(15)  // (diag) outputs: scalar loads
  if (lane < SE3D_NFU) {
    double *velk = Q.vel + (sl * SE3D_NFU + lane) * 9;
    if (measured) {
      const double q0 = L.xc[col][0], q1 = L.xc[col][1], q2 = L.xc[col][2];
      const double jf[3] = {q0 * height + root[0], q1 * height + root[1], q2 * height + root[2]};
      se3d_keypoint3d &kf = pf.keypoints[lane];
      for (int a = 0; a < 3; ++a) kf.joint[a] = jf[a];
      kf.score = fmaxf(0.10f, kscore);
      double c6[6];
      if (use_marg) for (int a = 0; a < 6; ++a) c6[a] = mc[a] * height * height;
      else { c6[0] = c6[3] = c6[5] = 0.10 * 0.10; c6[1] = c6[2] = c6[4] = 0.0; }
      if (lane == FU_MidHip) for (int a = 0; a < 6; ++a) c6[a] *= (100.0 * 100.0);
      for (int a = 0; a < 6; ++a) kf.cov[a] = c6[a];
      double jp[3] = {jf[0], jf[1], jf[2]};
      if (use_vel) {
        const double pv[3] = {x0[0] * hprev + rprev[0], x0[1] * hprev + rprev[1], x0[2] * hprev + rprev[2]};
        for (int a = 0; a < 3; ++a) velk[fslot * 3 + a] = (jf[a] - pv[a]) / (t - gtprev);
        for (int a = 0; a < 3; ++a) {
          const double acc = ((0.0 + velk[0 * 3 + a]) + velk[1 * 3 + a]) + velk[2 * 3 + a];
          jp[a] += acc / 3.0 * pred_dt;
        }
      }
      se3d_keypoint3d &kpred = pp.keypoints[lane];
      for (int a = 0; a < 3; ++a) kpred.joint[a] = jp[a];
      kpred.score = kf.score;
      for (int a = 0; a < 6; ++a) kpred.cov[a] = c6[a];
      kpred.cov[0] += 0.12 * 0.12; kpred.cov[3] += 0.12 * 0.12; kpred.cov[5] += 0.12 * 0.12;  // g_predNoiseSigma (PP:831)
      for (int a = 0; a < 3; ++a) Q.prev[(sl * SE3D_NFU + lane) * 3 + a] = L.xc[col][a];  // prevEstimate.swap(result)
    } else if ((hasprev >> lane) & 1u) {
      for (int a = 0; a < 9; ++a) velk[a] = 0.0;  // joint vanished: reset its velocity buffer (PP:490-493)
    }
  }
  if (lane == 0) {
    Q.has[sl] = measmask;
    Q.trk_t_prev[sl] = t;
    Q.height_prev[sl] = height;
    for (int a = 0; a < 3; ++a) Q.root_prev[sl * 3 + a] = root[a];
    const int no = Q.num_obs[sl] + 1;
    Q.num_obs[sl] = no;
    Q.pub[(size_t)s * H + i] = (no > 10) ? (Q.direct_out ? (uint8_t)pubv : 1) : 0;  // g_min_num_obs_track (PP:845-848)
    if (pubv) { Q.iters[(size_t)s * H + pubv - 1] = iterations; Q.obj[(size_t)s * H + pubv - 1] = obj_final; } else { Q.iters_det[(size_t)s * H + i] = iterations; Q.obj_det[(size_t)s * H + i] = obj_final; }
  }
  LM_STAMP(6)  // outputs
#ifdef SE3D_LM_STAMPS
  if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) { atomicAdd(&g_lm_stamps[8], 1ull); atomicAdd(&g_lm_stamps[9], (unsigned long long)iterations); }
#endif
}
template <int MODE>
__global__ __launch_bounds__(64) void se3d_k_lm_refine(Se3dStage2 Q) {
  extern __shared__ __align__(16) unsigned char s2_smem[];
  using LdsT = typename std::conditional<MODE != 0, LmSchurLds, LmLds>::type;
  if (Q.shard_world > 1 && (int)blockIdx.x % Q.shard_world != Q.shard_rank) return;  // hypothesis shard: another rank's detection
  s2_lm_body<MODE>(Q, blockIdx.x, blockIdx.y + Q.seq0, threadIdx.x, *reinterpret_cast<LdsT *>(s2_smem));
}

// =============================================================================================================
// row L + publishing — one workgroup per sequence (PP:845-852, 867-903, 909-910).
// =============================================================================================================
__device__ __forceinline__ void s2_track_update_body(const Se3dStage2 &Q, int s) {  // blockDim.x == 256
  const int tid = threadIdx.x;
  const int T = Q.T, H = Q.H;
  __shared__ int wsum[4];
  __shared__ int tmp[SE3D_MUNKRES_MAX];
  __shared__ int alive[SE3D_MUNKRES_MAX];
  __shared__ int srcdet[SE3D_MUNKRES_MAX];
  const int nd = min(Q.n_det[s], H);
  if (nd == 0) return;  // handled entirely in se3d_k_track_assoc
  const double t = Q.t[s];
  // publish in detection order
  int n_out;
  const bool pb = (tid < nd) && Q.pub[(size_t)s * H + tid];
  const int pos = s2_block_excl_scan(pb, wsum, n_out);
  if (pb) srcdet[pos] = tid;
  __syncthreads();
  const int words = (int)(sizeof(se3d_person3d) / 4);
  for (int idx = tid; idx < n_out * words; idx += 256) {
    const int p = idx / words, w = idx - p * words;
    const size_t src = (size_t)s * H + srcdet[p], dst = (size_t)s * H + p;
    reinterpret_cast<uint32_t *>(&Q.fused[dst])[w] = reinterpret_cast<const uint32_t *>(&Q.fused_det[src])[w];
    reinterpret_cast<uint32_t *>(&Q.pred[dst])[w] = reinterpret_cast<const uint32_t *>(&Q.pred_det[src])[w];
  }
  if (tid < n_out) { Q.iters[(size_t)s * H + tid] = Q.iters_det[(size_t)s * H + srcdet[tid]]; Q.obj[(size_t)s * H + tid] = Q.obj_det[(size_t)s * H + srcdet[tid]]; }
  __syncthreads();
  s2_remove_old(Q, s, t, wsum, tmp);
  // merge overlapping tracks (PP:870-903): sequential in i, all later j in parallel (track i is not modified by a merge)
  const int nt = Q.n_tracks[s];
  int *order = Q.order + (size_t)s * T;
  for (int k = tid; k < SE3D_MUNKRES_MAX; k += 256) alive[k] = (k < nt) ? 1 : 0;
  __syncthreads();
  for (int i = 0; i < nt; ++i) {
    if (!alive[i]) continue;
    const int j = tid;
    if (j > i && j < nt && alive[j]) {
      const size_t si = (size_t)s * T + order[i], sj = (size_t)s * T + order[j];
      const int slot_j = order[j];
      const double dd = ((Q.near[si * 4 + (slot_j >> 6)] >> (slot_j & 63)) & 1ull) ? 0.0 : 1.0;  // se3d_k_track_near
      if (dd < 0.20) {
        alive[j] = 0;
        Q.used[sj] = 0;
        const uint32_t idj = (uint32_t)Q.id[sj], idi = (uint32_t)Q.id[si];
        for (int p = 0; p < n_out; ++p)  // re-assign id in the persons messages (PP:893-898)
          if (Q.fused[(size_t)s * H + p].id == idj) { Q.fused[(size_t)s * H + p].id = idi; Q.pred[(size_t)s * H + p].id = idi; }
      }
    }
    __syncthreads();
  }
  int total;
  const bool kp = (tid < nt) && alive[tid];
  const int slotv = (tid < nt) ? order[tid] : -1;
  const int np = s2_block_excl_scan(kp, wsum, total);
  if (kp) tmp[np] = slotv;
  __syncthreads();
  if (tid < total) order[tid] = tmp[tid];
  if (tid == 0) {
    Q.n_tracks[s] = total;
    Q.n_out[s] = n_out;
    Q.t_prev[s] = t;
    Q.frame_nr[s] += 1;
  }
}

__global__ __launch_bounds__(256) void se3d_k_track_update(Se3dStage2 Q) { s2_track_update_body(Q, blockIdx.x + Q.seq0); }

// =============================================================================================================
// Fused multi-step walk: one workgroup per sequence runs track association -> LM refinement -> track lifecycle for all time
// steps of its sequence inside one launch.  The per-step kernels above are launch- and tail-latency bound when many short
// sequences are replayed from HBM (every launch waits for the slowest LM wavefront of all sequences); here each sequence
// advances at its own pace and the phases hand over through LDS / L1 with workgroup barriers only.
// Same device code as the per-step kernels (s2_*_body), so results are identical.
// =============================================================================================================
__device__ __forceinline__ void s2_wave_mem_sync() {  // stores of this wavefront visible to its other lanes' loads
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// =============================================================================================================
// Three persons per wavefront ("trio") — the LM of the fused walk.
// A person's factor graph has at most 21 variables (fusion joints) and, with the reference's bone tables, its range factors form a
// forest: every joint has at most one bone towards the root side (c_tree_rank); the only joint with two candidate parents is the
// neck (belly-neck, or midhip-neck when the belly is not measured: mutually exclusive).  So a lane can own joint j's variable, its
// unary factor AND the range factor of its parent bone, which makes a person 21 lanes wide: lanes 21 g + j, g = 0..2, carry three
// persons through the same instruction stream.  The LM control flow (GTSAM 4.0.3, as in s2_lm_body) is a per-person state machine
// held in group-uniform lane registers; one trip of the loop is one tryLambda trial for every person that is still running.
// Same algebra as s2_lm_body<3> (forest elimination for solves and marginals); sums run in a different order (rounding only).
// =============================================================================================================
struct LmTrioGroup {
  double xc[3][SE3D_NFU];                     // current values, by joint
  double xn[3][SE3D_NFU], dl[3][SE3D_NFU];    // tentative values, solution of the damped system; after the loop: marginal blocks
  double hs[3][SE3D_NFU];                     // parent-bone Jacobian direction / sigma of joint j (d = x[k2] - x[k1])
  double eb[SE3D_NFU];                        // parent-bone -whitened error, signed so that g_child += hs * eb, g_parent -= hs * eb
  double al[SE3D_NFU], be[SE3D_NFU];
};
struct LmTrioLds { LmTrioGroup g[3]; double red[2][64]; };
static_assert(offsetof(LmTrioGroup, dl) == offsetof(LmTrioGroup, xn) + sizeof(double) * 3 * SE3D_NFU, "marginal blocks alias xn + dl");

// sum over the 21 lanes of each group, identical on every lane of the group (fixed order)
__device__ __forceinline__ double s2_trio_sum(double v, double *red, int lane, int gb) {
  red[lane] = v;
  __builtin_amdgcn_wave_barrier();
  const double *r = red + gb;
  const double a0 = r[0] + r[1], a1 = r[2] + r[3], a2 = r[4] + r[5], a3 = r[6] + r[7], a4 = r[8] + r[9];
  const double a5 = r[10] + r[11], a6 = r[12] + r[13], a7 = r[14] + r[15], a8 = r[16] + r[17], a9 = r[18] + r[19];
  const double b0 = a0 + a1, b1 = a2 + a3, b2 = a4 + a5, b3 = a6 + a7, b4 = (a8 + a9) + r[20];
  const double tot = ((b0 + b1) + (b2 + b3)) + b4;
  __builtin_amdgcn_wave_barrier();
  return tot;
}

// 1/x and 1/sqrt(x) from the hardware seeds + two Newton steps (<= 1 ulp): the full-precision expansions of '/' and sqrt() are
// 10-20 dependent instructions each and sit on the critical path of every LM trial of a wavefront that walks its sequence alone
__device__ __forceinline__ double s2_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
  y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
  return y;
}
__device__ __forceinline__ double s2_rsq(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = __builtin_fma(__builtin_fma(-x * y, y, 1.0), 0.5 * y, y);
  y = __builtin_fma(__builtin_fma(-x * y, y, 1.0), 0.5 * y, y);
  return y;
}
__device__ __forceinline__ void s2_sym3_inv_fast(const double a[6], double o[6]) {
  const double c00 = a[3] * a[5] - a[4] * a[4], c01 = -(a[1] * a[5] - a[4] * a[2]), c02 = a[1] * a[4] - a[3] * a[2];
  const double det = a[0] * c00 + a[1] * c01 + a[2] * c02, id = s2_rcp(det);
  o[0] = c00 * id; o[1] = c01 * id; o[2] = c02 * id;
  o[3] = (a[0] * a[5] - a[2] * a[2]) * id; o[4] = -(a[0] * a[4] - a[1] * a[2]) * id; o[5] = (a[0] * a[3] - a[1] * a[1]) * id;
}

// two sums at once (one LDS round trip)
__device__ __forceinline__ void s2_trio_sum2(double &v0, double &v1, double (*red)[64], int lane, int gb) {
  red[0][lane] = v0; red[1][lane] = v1;
  __builtin_amdgcn_wave_barrier();
  double tot[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const double *r = red[q] + gb;
    const double a0 = r[0] + r[1], a1 = r[2] + r[3], a2 = r[4] + r[5], a3 = r[6] + r[7], a4 = r[8] + r[9];
    const double a5 = r[10] + r[11], a6 = r[12] + r[13], a7 = r[14] + r[15], a8 = r[16] + r[17], a9 = r[18] + r[19];
    const double b0 = a0 + a1, b1 = a2 + a3, b2 = a4 + a5, b3 = a6 + a7, b4 = (a8 + a9) + r[20];
    tot[q] = ((b0 + b1) + (b2 + b3)) + b4;
  }
  __builtin_amdgcn_wave_barrier();
  v0 = tot[0]; v1 = tot[1];
}

// upper-triangular whitener R (R0 R1 R2; 0 R4 R5; 0 0 R8) times a 3-vector
__device__ __forceinline__ void s2_trio_Rmul(const double R[6], const double v[3], double o[3]) {
  o[0] = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
  o[1] = R[3] * v[1] + R[4] * v[2];
  o[2] = R[5] * v[2];
}

// s, i, have: sequence (local index), detection and validity of the person of this lane's group (uniform within a group)
__device__ __forceinline__ void s2_lm_trio_body(const Se3dStage2 &Q, int s, int i, bool have, int lane, LmTrioLds &W) {
  const int H = Q.H, T = Q.T;
  const int g3 = lane / SE3D_NFU;
  const int g = g3 > 2 ? 2 : g3, j = lane - SE3D_NFU * g3, gb = SE3D_NFU * g;   // lane 63: no group
  LmTrioGroup &L = W.g[g];
  have = have && g3 < 3;   // the caller hands out detections i < min(n_det, H) only
  const int slot = have ? Q.det_slot[(size_t)s * H + i] : -1;
  const int pubv = (have && Q.direct_out) ? (int)Q.pub[(size_t)s * H + i] : 0;   // fused walk: 1 + output slot of a person that will be published
  have = have && slot >= 0;
  const size_t sl = have ? (size_t)s * T + slot : 0;
  const se3d_person3d &person = Q.dets[have ? (size_t)s * H + i : 0];
  const bool simple = Q.pose_method == SE3D_POSE_SIMPLE;
  const double t = Q.t[have ? s : 0];
  const unsigned gmask = 0x1FFFFFu;
  LM_STAMP_INIT

  // ---- root / neck / height (PP:631-668), identical on every lane of the group; measurements: lane 21 g + k <-> fusion joint k
  // of person g (PP:690-737).  Everything the branches below may need is fetched up front (one memory round trip): the four
  // keypoints root and neck derive from, this lane's own keypoint, the track's previous height / root.
  double root[3] = {0, 0, 0}, neck[3] = {0, 0, 0};
  float root_score = 0.f, neck_score = 0.f;
  double height = 1.0, hprev = 1.0, rprev[3] = {0, 0, 0};
  bool has_root = false, measured = false, first_obs = false;
  double m[3] = {0, 0, 0}, cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, xprev[3] = {0, 0, 0};
  float kscore = 0.f;
  if (have) {
    const int ia = simple ? FU_LHip : FU_MidHip, ib = simple ? FU_RHip : FU_MidHip, ic = simple ? FU_LShoulder : FU_Neck, id = simple ? FU_RShoulder : FU_Neck;
    const se3d_keypoint3d &ka = person.keypoints[ia], &kb = person.keypoints[ib], &kc = person.keypoints[ic], &kd = person.keypoints[id];
    const se3d_keypoint3d &kown = person.keypoints[j];
    double ja[3], jb[3], jc[3], jd[3], jo[3], ca6[6], cb6[6], co6[6];
    for (int a = 0; a < 3; ++a) { ja[a] = ka.joint[a]; jb[a] = kb.joint[a]; jc[a] = kc.joint[a]; jd[a] = kd.joint[a]; jo[a] = kown.joint[a]; }
    const float sa = ka.score, sb = kb.score, sc = kc.score, sd = kd.score, so = kown.score;
    // covariances: own keypoint; MidHip lane: of the two hips (simple) / own; Neck lane (simple): of the two shoulders
    const bool hip_lane = j == FU_MidHip, neck_lane = simple && j == FU_Neck;
    const se3d_keypoint3d &kca = hip_lane ? ka : kc, &kcb = hip_lane ? kb : kd;
    for (int q = 0; q < 6; ++q) { co6[q] = kown.cov[q]; ca6[q] = kca.cov[q]; cb6[q] = kcb.cov[q]; }
    const double hp_old = Q.height_prev[sl];
    double rp_old[3];
    for (int a = 0; a < 3; ++a) { rp_old[a] = Q.root_prev[sl * 3 + a]; xprev[a] = Q.prev[(sl * SE3D_NFU + j) * 3 + a]; }

    if (!simple) {
      for (int a = 0; a < 3; ++a) { root[a] = ja[a]; neck[a] = jc[a]; }
      root_score = sa;
      neck_score = sc;
    } else {
      if (sa > 0.0f && sb > 0.0f) {
        for (int a = 0; a < 3; ++a) root[a] = (ja[a] + jb[a]) / 2.0;
        root_score = (sa + sb) / 2.0f;
      }
      if (sc > 0.0f && sd > 0.0f) {
        for (int a = 0; a < 3; ++a) neck[a] = (jc[a] + jd[a]) / 2.0;
        neck_score = (sc + sd) / 2.0f;
      }
    }
    has_root = root_score > 0.10f;
    if (has_root && Q.norm_height) {
      if (neck_score > 0.10f) {
        const double dx = neck[0] - root[0], dy = neck[1] - root[1], dz = neck[2] - root[2];
        height = sqrt(dx * dx + dy * dy + dz * dz);
      } else height = 0.60;
    }
    // 1 / height and 1 / height^2 once: the branches below scale nine covariance entries and three coordinates each, and a wavefront runs
    // all three branches (hip, neck, other lanes) — 42 fp64 IEEE divisions (~40 instructions each) per pass, a fifth of the walk's
    // instructions, on the critical path of every LM set-up.  The products differ from the reference's quotients by <= 1 ulp of a double
    // (exactly equal when heights are not normalised: height = 1); the trio LM is compared with the per-step kernels at 1e-9 m anyway.
    const double ih = 1.0 / height, ih2 = 1.0 / (height * height);
    const bool first = hp_old < 0.0;  // PP:699-702
    first_obs = first;
    hprev = first ? height : hp_old;
    for (int a = 0; a < 3; ++a) rprev[a] = first ? root[a] : rp_old[a];
    if (first && j == 0) { Q.height_prev[sl] = height; for (int a = 0; a < 3; ++a) Q.root_prev[sl * 3 + a] = root[a]; }

    if (hip_lane) {
      if (has_root) {
        measured = true;
        if (!simple) s2_cov6_to_mat(co6, cov);
        else {
          double cl[9], cr[9];
          s2_cov6_to_mat(ca6, cl);
          s2_cov6_to_mat(cb6, cr);
          for (int q = 0; q < 9; ++q) cov[q] = (cl[q] + cr[q]) / 2.0;
        }
        for (int q = 0; q < 9; ++q) cov[q] = cov[q] * ih2 * (1.0 / (100.0 * 100.0));
        kscore = root_score;
      }
    } else if (neck_lane && neck_score > 0.10f) {
      measured = true;
      double cl[9], cr[9];
      s2_cov6_to_mat(ca6, cl);
      s2_cov6_to_mat(cb6, cr);
      for (int q = 0; q < 9; ++q) cov[q] = (cl[q] + cr[q]) / 2.0 * ih2;
      for (int a = 0; a < 3; ++a) m[a] = (neck[a] - root[a]) * ih;
      kscore = neck_score;
    } else if (so > 0.10f) {
      measured = true;
      double c9[9];
      s2_cov6_to_mat(co6, c9);
      for (int q = 0; q < 9; ++q) cov[q] = c9[q] * ih2;
      for (int a = 0; a < 3; ++a) m[a] = (jo[a] - root[a]) * ih;
      kscore = so;
    }
  }
  const uint32_t measmask = (uint32_t)(__ballot(measured) >> gb) & gmask;
  const bool valid = have && measmask != 0u;  // PP:739-741: a person without measured joints is skipped
  const uint32_t hasprev = valid ? Q.has[sl] : 0u;
  const bool use_vel = measured && ((hasprev >> j) & 1u);
  bool wok = true;
  double R[6] = {0, 0, 0, 0, 0, 0}, x0[3] = {0, 0, 0}, x[3] = {0, 0, 0};
  if (measured) {
    double R9[9];
    wok = s2_make_whitener(cov, R9);
    R[0] = R9[0]; R[1] = R9[1]; R[2] = R9[2]; R[3] = R9[4]; R[4] = R9[5]; R[5] = R9[8];
    for (int a = 0; a < 3; ++a) {
      x0[a] = use_vel ? xprev[a] : m[a];  // setInitialState (PP:483-503)
      x[a] = x0[a];
      L.xc[a][j] = x[a];
    }
  }
  bool wh_ok = ((uint32_t)(__ballot(measured && !wok) >> gb) & gmask) == 0u;

  // ---- range factors (PP:384-481): lane 21 g + e evaluates bone e of the table; joint j then finds its parent bone and children ----
  const int variant = Q.norm_height ? 1 : 0;
  bool ract = false;
  {
    const Se3dBone bd = c_bones[0][j];
    ract = valid && ((measmask >> bd.k1) & 1u) && ((measmask >> bd.k2) & 1u) && (!bd.not_belly || !((measmask >> FU_Belly) & 1u));
  }
  const uint32_t rbg = (uint32_t)(__ballot(ract) >> gb) & gmask;
  int pe = -1, parent = -1, npar = 0;
  bool child_is_k2 = true;
  unsigned children = 0u;
  double bL = 0.0, bS = 1.0;
#pragma unroll
  for (int e = 0; e < 21; ++e) {
    const int k1 = c_bones[0][e].k1, k2 = c_bones[0][e].k2;  // the two tables share the bone ends
    const bool k1_child = c_tree_rank[k1] > c_tree_rank[k2];
    const int ch = k1_child ? k1 : k2, pa = k1_child ? k2 : k1;
    const bool on = (rbg >> e) & 1u;
    if (on && j == ch) { pe = e; parent = pa; ++npar; child_is_k2 = !k1_child; bL = c_bones[variant][e].L; bS = c_bones[variant][e].sigma; }
    if (on && j == pa) children |= 1u << ch;
  }
  bS *= (Q.norm_height ? 2.0 : 1.0);  // g_limbLSigmaFactor (PP:934-937)
  const double invS = 1.0 / bS;
  if (!measured) { pe = -1; parent = -1; children = 0u; }
  if (((uint32_t)(__ballot(measured && (npar > 1 || __popc(children) > 4)) >> gb) & gmask) != 0u) {  // impossible with the bone tables above
    wh_ok = false;
    if (valid && j == 0) atomicOr(&Q.flags2[s], 8u);
  }
  int level = 0;
  {
    int cur = parent;
    while (__ballot(cur >= 0)) {
      if (cur >= 0) ++level;
      const int nxt = __shfl(parent, gb + (cur >= 0 ? cur : 0));
      cur = cur >= 0 ? nxt : -1;
    }
  }
  int maxlev = 0;
  for (int lv = SE3D_NFU - 1; lv > 0; --lv)
    if (__ballot(measured && level == lv)) { maxlev = lv; break; }
  const bool edge = pe >= 0;

  // children of this joint (at most four with the bone tables), visited two at a time: their LDS rows are fetched together
  int chv[4];
  const int nch = __popc(children);
  {
    unsigned cm = children;
#pragma unroll
    for (int q = 0; q < 4; ++q) { chv[q] = cm ? __ffs(cm) - 1 : j; cm &= cm - 1; }
  }
  const int nbatch = __ballot(nch > 2) != 0ull ? 4 : 2;
  unsigned wide_levels = 0u;  // bit l: some joint of level l has more than two children
  for (int lv = 0; lv < maxlev; ++lv)
    if (__ballot(measured && level == lv && nch > 2)) wide_levels |= 1u << lv;
  __builtin_amdgcn_wave_barrier();

  // Whitened errors of this lane's factors at xv (own joint; parent joint from X): unary part eu, parent-bone part ee.  The evaluation
  // leaves what a linearisation at the same point needs in registers (whitened unary residual, bone vector, length, 1 / length): LM
  // only ever re-linearises at a point whose error it has just evaluated (the initial values, an accepted step).
  double ev_w[3] = {0, 0, 0}, ev_d[3] = {0, 0, 0}, ev_r = 0.0, ev_ir = 0.0;
  auto evaluate = [&](const double xv[3], const double (*X)[SE3D_NFU], double &eu, double &ee) {
    eu = 0.0; ee = 0.0;
    if (measured) {
      const double r3[3] = {xv[0] - m[0], xv[1] - m[1], xv[2] - m[2]};
      s2_trio_Rmul(R, r3, ev_w);
      eu = 0.5 * ev_w[0] * ev_w[0] + 0.5 * ev_w[1] * ev_w[1] + 0.5 * ev_w[2] * ev_w[2];
      if (edge) {
        const double px = X[0][parent], py = X[1][parent], pz = X[2][parent];
        ev_d[0] = child_is_k2 ? xv[0] - px : px - xv[0]; ev_d[1] = child_is_k2 ? xv[1] - py : py - xv[1]; ev_d[2] = child_is_k2 ? xv[2] - pz : pz - xv[2];
        const double r2 = ev_d[0] * ev_d[0] + ev_d[1] * ev_d[1] + ev_d[2] * ev_d[2];
        ev_ir = r2 > 0.0 ? s2_rsq(r2) : 0.0;
        ev_r = r2 * ev_ir;
        const double w = (ev_r - bL) * invS;
        ee = 0.5 * w * w;
      }
    }
  };
  // Linearisation at the evaluated point, for the groups with `on`: parent-bone direction / sigma hsv and rhs ebu (LDS copies for the
  // parent lane), gradient gv = J^T b and block diagonal A0 = (J^T J)_jj of this joint (own unary and parent bone, children's bones).
  double hsv[3] = {0, 0, 0}, ebu = 0.0, A0[6] = {0, 0, 0, 0, 0, 0}, gv[3] = {0, 0, 0};
  auto relinearize = [&](bool on) {
    on = on && measured;
    if (on) {
      gv[0] = -(R[0] * ev_w[0]); gv[1] = -(R[1] * ev_w[0] + R[3] * ev_w[1]); gv[2] = -(R[2] * ev_w[0] + R[4] * ev_w[1] + R[5] * ev_w[2]);
      A0[0] = R[0] * R[0]; A0[1] = R[0] * R[1]; A0[2] = R[0] * R[2];   // R^T R
      A0[3] = R[1] * R[1] + R[3] * R[3]; A0[4] = R[1] * R[2] + R[3] * R[4]; A0[5] = R[2] * R[2] + R[4] * R[4] + R[5] * R[5];
      if (edge) {
        double h0 = 1.0, h1 = 1.0, h2 = 1.0;
        if (fabs(ev_r) > 1e-10) { h0 = ev_d[0] * ev_ir; h1 = ev_d[1] * ev_ir; h2 = ev_d[2] * ev_ir; }
        hsv[0] = h0 * invS; hsv[1] = h1 * invS; hsv[2] = h2 * invS;
        ebu = -(ev_r - bL) * invS;
        const double sb = child_is_k2 ? ebu : -ebu;
        L.hs[0][j] = hsv[0]; L.hs[1][j] = hsv[1]; L.hs[2][j] = hsv[2];
        L.eb[j] = sb;
        gv[0] += hsv[0] * sb; gv[1] += hsv[1] * sb; gv[2] += hsv[2] * sb;
        A0[0] += hsv[0] * hsv[0]; A0[1] += hsv[0] * hsv[1]; A0[2] += hsv[0] * hsv[2];
        A0[3] += hsv[1] * hsv[1]; A0[4] += hsv[1] * hsv[2]; A0[5] += hsv[2] * hsv[2];
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (on) {
      for (int q0 = 0; q0 < nbatch; q0 += 2) {
        double c0[2], c1[2], c2[2], cb[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) { const int c = chv[q0 + q]; c0[q] = L.hs[0][c]; c1[q] = L.hs[1][c]; c2[q] = L.hs[2][c]; cb[q] = L.eb[c]; }
#pragma unroll
        for (int q = 0; q < 2; ++q)
          if (q0 + q < nch) {
            const double h0 = c0[q], h1 = c1[q], h2 = c2[q], sb = cb[q];
            gv[0] -= h0 * sb; gv[1] -= h1 * sb; gv[2] -= h2 * sb;
            A0[0] += h0 * h0; A0[1] += h0 * h1; A0[2] += h0 * h2; A0[3] += h1 * h1; A0[4] += h1 * h2; A0[5] += h2 * h2;
          }
      }
    }
  };

  // ---- LevenbergMarquardtOptimizer(graph, prevEstimate).optimize(), GTSAM 4.0.3 defaults ----
  // GTSAM's two nested loops (iterate / tryLambda) as one loop of trials: `tr` counts the trials of the current iteration, an accepted
  // step (or giving up on lambda) ends the iteration and runs its convergence checks.
  int iterations = 0, tr = 0, outer = 0;
  double error = 0.0, lambda = 1e-5, currentError = 0.0;
  bool running = false;
  {
    double eu, ee;
    evaluate(x, L.xc, eu, ee);
    error = s2_trio_sum(eu + ee, W.red[0], lane, gb);
    running = valid && wh_ok && !(error <= 0.0);
    currentError = error;
    if (__ballot(running)) relinearize(running);
  }
  LM_STAMP(0)  // setup + initial error + first linearisation
  bool pending = false;
  while (__ballot(running)) {
    if (__ballot(pending)) { relinearize(pending); pending = false; }   // at the accepted values, from their evaluation
    LM_STAMP(1)  // linearise + gradient
    // ---- tryLambda: delta = (J^T J + lambda I)^-1 J^T b by forest elimination ----
    const bool act = running && measured;
    double u[3] = {0, 0, 0}, z[3] = {0, 0, 0}, d[3] = {0, 0, 0}, dp[3] = {0, 0, 0};
    {
      double D[6] = {A0[0] + lambda, A0[1], A0[2], A0[3] + lambda, A0[4], A0[5] + lambda};
      double gg[3] = {gv[0], gv[1], gv[2]};
      double Di[6] = {0, 0, 0, 0, 0, 0};
      for (int lv = maxlev; lv >= 1; --lv) {  // leaves -> roots
        if (act && level == lv) {
          s2_sym3_inv_fast(D, Di);
          s2_sym3_mul(Di, hsv, u);
          s2_sym3_mul(Di, gg, z);
          L.al[j] = hsv[0] * u[0] + hsv[1] * u[1] + hsv[2] * u[2];
          L.be[j] = hsv[0] * z[0] + hsv[1] * z[1] + hsv[2] * z[2];
        }
        __builtin_amdgcn_wave_barrier();
        if (act && level == lv - 1) {
          for (int q0 = 0; q0 < (((wide_levels >> (lv - 1)) & 1u) ? 4 : 2); q0 += 2) {
            double c0[2], c1[2], c2[2], ca[2], cb[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) { const int c = chv[q0 + q]; c0[q] = L.hs[0][c]; c1[q] = L.hs[1][c]; c2[q] = L.hs[2][c]; ca[q] = L.al[c]; cb[q] = L.be[c]; }
#pragma unroll
            for (int q = 0; q < 2; ++q)
              if (q0 + q < nch) {
                const double h0 = c0[q], h1 = c1[q], h2 = c2[q], a_ = ca[q], b_ = cb[q];
                D[0] -= a_ * h0 * h0; D[1] -= a_ * h0 * h1; D[2] -= a_ * h0 * h2; D[3] -= a_ * h1 * h1; D[4] -= a_ * h1 * h2; D[5] -= a_ * h2 * h2;
                gg[0] += b_ * h0; gg[1] += b_ * h1; gg[2] += b_ * h2;
              }
          }
        }
      }
      if (act && level == 0) {
        s2_sym3_inv_fast(D, Di);
        s2_sym3_mul(Di, gg, d);
        for (int a = 0; a < 3; ++a) L.dl[a][j] = d[a];
      }
    }
    for (int lv = 1; lv <= maxlev; ++lv) {  // roots -> leaves
      __builtin_amdgcn_wave_barrier();
      if (act && level == lv) {
        for (int a = 0; a < 3; ++a) dp[a] = L.dl[a][parent];
        const double gam = hsv[0] * dp[0] + hsv[1] * dp[1] + hsv[2] * dp[2];
        for (int a = 0; a < 3; ++a) { d[a] = z[a] + u[a] * gam; L.dl[a][j] = d[a]; }
      }
    }
    LM_STAMP(2)  // solve
    // Retract, then both errors of the trial.  linear.error(delta) = 0.5 ||J delta - b||^2 over the undamped factors: a unary factor
    // is linear, so its linearised error at delta IS its error at the tentative values; only the bones differ.
    const double xn[3] = {x[0] + d[0], x[1] + d[1], x[2] + d[2]};
    double le = 0.0, newError = 0.0;
    if (act) {
      if (edge) {
        const double e0 = child_is_k2 ? d[0] - dp[0] : dp[0] - d[0], e1 = child_is_k2 ? d[1] - dp[1] : dp[1] - d[1],
                     e2 = child_is_k2 ? d[2] - dp[2] : dp[2] - d[2];
        const double sv = hsv[0] * e0 + hsv[1] * e1 + hsv[2] * e2 - ebu;
        le = 0.5 * sv * sv;
      }
      for (int a = 0; a < 3; ++a) L.xn[a][j] = xn[a];
    }
    __builtin_amdgcn_wave_barrier();  // xn visible to the children
    if (act) {
      double eu, ee;
      evaluate(xn, L.xn, eu, ee);
      le += eu;
      newError = eu + ee;
    }
    s2_trio_sum2(le, newError, W.red, lane, gb);
    bool accepted = false;
    if (running) {
      const double linearizedCostChange = error - le;
      bool step_ok = false, stop = false;
      if (linearizedCostChange >= 0) {
        const double costChange = error - newError;
        if (linearizedCostChange > 1e-20) step_ok = costChange > 1e-3 * linearizedCostChange;  // modelFidelity > minModelFidelity
        if (fabs(costChange) < 1e-5 * error) stop = true;
      }
      bool end_outer = false;
      if (step_ok) {  // decreaseLambda (useFixedLambdaFactor)
        if (measured)
          for (int a = 0; a < 3; ++a) { x[a] = xn[a]; L.xc[a][j] = xn[a]; }
        error = newError;
        lambda /= 10.0;
        if (lambda < 0.0) lambda = 0.0;
        ++iterations;
        end_outer = true;
      } else if (!stop) {  // increaseLambda
        lambda *= 10.0;
        if (lambda >= 1e5 || ++tr >= 64) end_outer = true;
      } else {
        end_outer = true;
      }
      if (end_outer) {  // checks at the end of an iteration; without an accepted step the error is unchanged and they stop the loop
        const double absoluteDecrease = currentError - error;
        ++outer;
        if (iterations >= 100 || error <= 0.0 || absoluteDecrease <= 1e-5 * currentError /* relativeDecrease <= relativeErrorTol */ ||
            absoluteDecrease <= 1e-5 || !isfinite(currentError) || outer >= 1000)
          running = false;
        currentError = error;
        tr = 0;
        accepted = step_ok && running;
      }
    }
    LM_STAMP(3)  // errors, sums, control
    pending = accepted;
    __builtin_amdgcn_wave_barrier();
#ifdef SE3D_LM_STAMPS
    if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) atomicAdd(&g_lm_stamps[9], 1ull);
#endif
  }
  if (valid && !wh_ok && measured) { x[0] = m[0]; x[1] = m[1]; x[2] = m[2]; }  // optimisation failed: output the measurements (PP:751-758)
  if (valid && j == 0) {   // graph.error(result) (se3d_fused_out.lm_objective), stored here so that it is not carried through the output phase
    const double o = wh_ok ? error : -1.0;
    if (pubv) Q.obj[(size_t)s * H + pubv - 1] = o; else Q.obj_det[(size_t)s * H + i] = o;
  }

  // ---- Marginals(graph, result): diagonal 3x3 blocks of (J^T J)^-1 at the optimum (PP:760-793), top-down over the forest ----
  const bool use_marg = valid && wh_ok;
  double mc[6] = {0, 0, 0, 0, 0, 0};
  if (__ballot(use_marg)) {
    const bool act = use_marg && measured;
    double *MC = &L.xn[0][0];  // [6][21] over xn + dl
    {
      double eu, ee;
      evaluate(x, L.xc, eu, ee);   // the last trial may have been rejected: evaluate at the result
      relinearize(use_marg);       // A0 = block diagonal of J^T J
    }
    double D[6] = {A0[0], A0[1], A0[2], A0[3], A0[4], A0[5]}, Di[6] = {0, 0, 0, 0, 0, 0}, u[3] = {0, 0, 0};
    for (int lv = maxlev; lv >= 1; --lv) {
      if (act && level == lv) {
        s2_sym3_inv_fast(D, Di);
        s2_sym3_mul(Di, hsv, u);
        L.al[j] = hsv[0] * u[0] + hsv[1] * u[1] + hsv[2] * u[2];
      }
      __builtin_amdgcn_wave_barrier();
      if (act && level == lv - 1) {
        for (int q0 = 0; q0 < (((wide_levels >> (lv - 1)) & 1u) ? 4 : 2); q0 += 2) {
          double c0[2], c1[2], c2[2], ca[2];
#pragma unroll
          for (int q = 0; q < 2; ++q) { const int c = chv[q0 + q]; c0[q] = L.hs[0][c]; c1[q] = L.hs[1][c]; c2[q] = L.hs[2][c]; ca[q] = L.al[c]; }
#pragma unroll
          for (int q = 0; q < 2; ++q)
            if (q0 + q < nch) {
              const double h0 = c0[q], h1 = c1[q], h2 = c2[q], a_ = ca[q];
              D[0] -= a_ * h0 * h0; D[1] -= a_ * h0 * h1; D[2] -= a_ * h0 * h2; D[3] -= a_ * h1 * h1; D[4] -= a_ * h1 * h2; D[5] -= a_ * h2 * h2;
            }
        }
      }
    }
    if (act && level == 0) {
      s2_sym3_inv_fast(D, Di);
      for (int q = 0; q < 6; ++q) { mc[q] = Di[q]; MC[q * SE3D_NFU + j] = Di[q]; }
    }
    for (int lv = 1; lv <= maxlev; ++lv) {
      __builtin_amdgcn_wave_barrier();
      if (act && level == lv) {
        double sp[6], sh[3];
        for (int q = 0; q < 6; ++q) sp[q] = MC[q * SE3D_NFU + parent];
        s2_sym3_mul(sp, hsv, sh);
        const double gam = hsv[0] * sh[0] + hsv[1] * sh[1] + hsv[2] * sh[2];
        mc[0] = Di[0] + gam * u[0] * u[0]; mc[1] = Di[1] + gam * u[0] * u[1]; mc[2] = Di[2] + gam * u[0] * u[2];
        mc[3] = Di[3] + gam * u[1] * u[1]; mc[4] = Di[4] + gam * u[1] * u[2]; mc[5] = Di[5] + gam * u[2] * u[2];
        for (int q = 0; q < 6; ++q) MC[q * SE3D_NFU + j] = mc[q];
      }
    }
    __builtin_amdgcn_wave_barrier();
  }

  LM_STAMP(5)  // marginals
  // ---- outputs + track-state update (PP:769-843): every lane writes the whole keypoint of its joint in both persons ----
  if (valid) {
#ifndef SE3D_TRIO_CARRY
    // What only this phase needs — root, height, the previous root / height, the initial value, the score, the stamp — is derived AGAIN
    // from memory here (the detection and the track slot are unchanged, L2 / LDS hits) instead of being carried across the trial loop:
    // 13 doubles per lane that the loop never touches were spilled to scratch before every pass and reloaded after it.
    asm volatile("" ::: "memory");
    {
      const int ia = simple ? FU_LHip : FU_MidHip, ib = simple ? FU_RHip : FU_MidHip, ic = simple ? FU_LShoulder : FU_Neck, id = simple ? FU_RShoulder : FU_Neck;
      const se3d_keypoint3d &ka = person.keypoints[ia], &kb = person.keypoints[ib], &kc = person.keypoints[ic], &kd = person.keypoints[id];
      const float sa = ka.score, sb = kb.score, sc = kc.score, sd = kd.score, so = person.keypoints[j].score;
      double nk[3] = {0, 0, 0};
      float rs = 0.f, ns = 0.f;
      root[0] = root[1] = root[2] = 0.0;
      if (!simple) {
        for (int a = 0; a < 3; ++a) { root[a] = ka.joint[a]; nk[a] = kc.joint[a]; }
        rs = sa; ns = sc;
      } else {
        if (sa > 0.0f && sb > 0.0f) {
          for (int a = 0; a < 3; ++a) root[a] = (ka.joint[a] + kb.joint[a]) / 2.0;
          rs = (sa + sb) / 2.0f;
        }
        if (sc > 0.0f && sd > 0.0f) {
          for (int a = 0; a < 3; ++a) nk[a] = (kc.joint[a] + kd.joint[a]) / 2.0;
          ns = (sc + sd) / 2.0f;
        }
      }
      height = 1.0;
      if (rs > 0.10f && Q.norm_height) {
        if (ns > 0.10f) {
          const double dx = nk[0] - root[0], dy = nk[1] - root[1], dz = nk[2] - root[2];
          height = sqrt(dx * dx + dy * dy + dz * dz);
        } else height = 0.60;
      }
      kscore = j == FU_MidHip ? rs : ((simple && j == FU_Neck) ? ns : so);
      // a first observation stored height / root into the slot before the loop: reading them back gives `first ? height : old` either way
      hprev = first_obs ? height : Q.height_prev[sl];
      for (int a = 0; a < 3; ++a) {
        rprev[a] = first_obs ? root[a] : Q.root_prev[sl * 3 + a];
        x0[a] = use_vel ? Q.prev[(sl * SE3D_NFU + j) * 3 + a] : m[a];
      }
    }
#endif
    const bool wr = !Q.direct_out || pubv != 0;                          // an unpublished person's payload is never read: not written
    se3d_person3d &pf = pubv ? Q.fused[(size_t)s * H + pubv - 1] : Q.fused_det[(size_t)s * H + i];
    se3d_person3d &pp = pubv ? Q.pred[(size_t)s * H + pubv - 1] : Q.pred_det[(size_t)s * H + i];
    const int track_id = Q.id[sl];
    if (j == 0 && wr) { pf.id = (uint32_t)track_id; pf.score = 0.f; pp.id = (uint32_t)track_id; pp.score = 0.f; }
    const double pred_dt = Q.u_valid ? Q.u_pred_dt : Q.pred_dt[s];
    const int fslot = Q.u_valid ? Q.u_fslot : Q.frame_nr[s] % 3;
    const double gtprev = Q.u_valid ? Q.u_t_prev : Q.t_prev[s];
    double *velk = Q.vel + (sl * SE3D_NFU + j) * 9;
    se3d_keypoint3d kf, kpred;
    for (int a = 0; a < 3; ++a) { kf.joint[a] = 0.0; kpred.joint[a] = 0.0; }
    for (int a = 0; a < 6; ++a) { kf.cov[a] = 0.0; kpred.cov[a] = 0.0; }
    kf.score = 0.f; kf._pad = 0.f; kpred.score = 0.f; kpred._pad = 0.f;
    if (measured) {
      const double jf[3] = {x[0] * height + root[0], x[1] * height + root[1], x[2] * height + root[2]};
      for (int a = 0; a < 3; ++a) kf.joint[a] = jf[a];
      kf.score = fmaxf(0.10f, kscore);
      double c6[6];
      if (use_marg) for (int a = 0; a < 6; ++a) c6[a] = mc[a] * height * height;
      else { c6[0] = c6[3] = c6[5] = 0.10 * 0.10; c6[1] = c6[2] = c6[4] = 0.0; }
      if (j == FU_MidHip) for (int a = 0; a < 6; ++a) c6[a] *= (100.0 * 100.0);
      for (int a = 0; a < 6; ++a) kf.cov[a] = c6[a];
      double jp[3] = {jf[0], jf[1], jf[2]};
      if (use_vel) {
        const double pv[3] = {x0[0] * hprev + rprev[0], x0[1] * hprev + rprev[1], x0[2] * hprev + rprev[2]};
        double vk[9];
        for (int a = 0; a < 9; ++a) vk[a] = velk[a];
        const double idt = 1.0 / (t - gtprev);   // one division for the three velocity components (<= 1 ulp from the three quotients)
        for (int a = 0; a < 3; ++a) {
          const double vnew = (jf[a] - pv[a]) * idt;
          velk[fslot * 3 + a] = vnew;
          vk[0 * 3 + a] = fslot == 0 ? vnew : vk[0 * 3 + a]; vk[1 * 3 + a] = fslot == 1 ? vnew : vk[1 * 3 + a]; vk[2 * 3 + a] = fslot == 2 ? vnew : vk[2 * 3 + a];
          const double acc = ((0.0 + vk[0 * 3 + a]) + vk[1 * 3 + a]) + vk[2 * 3 + a];
          jp[a] += acc * (1.0 / 3.0) * pred_dt;
        }
      }
      for (int a = 0; a < 3; ++a) kpred.joint[a] = jp[a];
      kpred.score = kf.score;
      for (int a = 0; a < 6; ++a) kpred.cov[a] = c6[a];
      kpred.cov[0] += 0.12 * 0.12; kpred.cov[3] += 0.12 * 0.12; kpred.cov[5] += 0.12 * 0.12;  // g_predNoiseSigma (PP:831)
      for (int a = 0; a < 3; ++a) Q.prev[(sl * SE3D_NFU + j) * 3 + a] = x[a];  // prevEstimate.swap(result)
    } else if ((hasprev >> j) & 1u) {
      for (int a = 0; a < 9; ++a) velk[a] = 0.0;  // joint vanished: reset its velocity buffer (PP:490-493)
    }
    if (wr) { pf.keypoints[j] = kf; pp.keypoints[j] = kpred; }
    if (j == 0) {
      Q.has[sl] = measmask;
      Q.trk_t_prev[sl] = t;
      Q.height_prev[sl] = height;
      for (int a = 0; a < 3; ++a) Q.root_prev[sl * 3 + a] = root[a];
      const int no = Q.num_obs[sl] + 1;
      Q.num_obs[sl] = no;
      Q.pub[(size_t)s * H + i] = (no > 10) ? (Q.direct_out ? (uint8_t)pubv : 1) : 0;  // g_min_num_obs_track (PP:845-848)
      if (pubv) Q.iters[(size_t)s * H + pubv - 1] = iterations; else Q.iters_det[(size_t)s * H + i] = iterations;
    }
  }
  LM_STAMP(6)  // outputs
#ifdef SE3D_LM_STAMPS
  if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) atomicAdd(&g_lm_stamps[8], 1ull);
#endif
}

// ---- single-wavefront forms of the per-sequence bookkeeping (T, H <= 64): same decisions, ballots instead of barriers ----
__device__ __forceinline__ int s2_nth_set_bit(unsigned long long m, int n) {  // index of the n-th (0-based) set bit, -1 if none
  for (int i = 0; i < n; ++i) m &= m - 1ull;
  return m ? __ffsll((long long)m) - 1 : -1;
}

__device__ __forceinline__ void s2_remove_old_wave(const Se3dStage2 &Q, int s, double t, int lane) {  // PP:191-211
  const int T = Q.T;
  const int nt = Q.n_tracks[s];
  int *order = Q.order + (size_t)s * T;
  int slot = -1;
  bool keep = false;
  if (lane < nt) {
    slot = order[lane];
    keep = !(t - Q.trk_t_prev[(size_t)s * T + slot] > 1.0);
    if (!keep) Q.used[(size_t)s * T + slot] = 0;
  }
  const unsigned long long km = __ballot(keep);
  if (keep) order[__popcll(km & ((1ull << lane) - 1ull))] = slot;
  if (lane == 0) Q.n_tracks[s] = __popcll(km);
}

// rows Y + A2 (PP:513-580) by one wavefront.  W: LDS scratch of nd * nt doubles.
__device__ __forceinline__ void s2_track_assoc_wave(const Se3dStage2 &Q, int s, int lane, double *W) {
  const int T = Q.T, H = Q.H, N = Q.N;
  const double t = Q.t[s];
  const int nd = min(Q.n_det[s], H);
  const int nt = Q.n_tracks[s];
  // feedback-delay moving average -> pred_delta_t (PP:513-526): the N delays are fetched by N lanes in one memory round trip (a loop of
  // loads on lane 0 was N round trips) and summed in camera order from the lanes' registers
  double fb_acc = 0.0;
  int fb_nv = 0;
  for (int base = 0; base < N; base += 64) {
    const float dl = (Q.fb_delay && base + lane < N) ? Q.fb_delay[(size_t)s * N + base + lane] : -1.0f;
    const int cnt = min(64, N - base);
    for (int i = 0; i < cnt; ++i) {
      const float d = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, dl), i));
      if (d > 0.0f) { fb_acc += (double)d; ++fb_nv; }
    }
  }
  if (lane == 0) {
    const double acc = fb_acc;
    const int nv = fb_nv;
    const double cur = nv > 0 ? acc / nv : 0.10;
    Q.fb_buf[s * 3 + (Q.frame_nr[s] % 3)] = cur;
    Q.pred_dt[s] = (0.0 + Q.fb_buf[s * 3 + 0] + Q.fb_buf[s * 3 + 1] + Q.fb_buf[s * 3 + 2]) / 3;
    Q.n_out[s] = 0;
  }
  if (lane < H) { Q.det_slot[(size_t)s * H + lane] = -1; Q.pub[(size_t)s * H + lane] = 0; }
  if (nd == 0) {  // PP:537-546
    s2_remove_old_wave(Q, s, t, lane);
    if (lane == 0) Q.t_prev[s] = t;
    return;
  }
  int *order = Q.order + (size_t)s * T;
  const double *C = Q.C + (size_t)s * H * T;
  int assign = -1;
  if (nt > 0) {
    if (lane < nd)
      for (int tk = 0; tk < nt; ++tk) W[lane + nd * tk] = C[lane + (size_t)nd * tk];
    const int rc = se3d_munkres_wave(W, nd, nd, nt, lane, assign);
    if (rc != 0 && lane == 0) atomicOr(&Q.flags2[s], 4u);
    if (lane < nd && assign >= 0 && C[lane + (size_t)nd * assign] > 5.0) assign = -1;  // PP:562-565
  }
  // new tracks for unassigned detections, in detection order (PP:570-580)
  const bool isnew = (lane < nd) && (nt == 0 || assign < 0);
  const unsigned long long nb = __ballot(isnew);
  const int p = __popcll(nb & ((1ull << lane) - 1ull)), total_new = __popcll(nb);
  const unsigned long long frb = __ballot((lane < T) && !Q.used[(size_t)s * T + lane]);
  const int total_free = __popcll(frb);
  const int next_id = Q.next_id[s];
  int my_slot = -1;   // track slot of this lane's detection (what it writes to det_slot)
  if (lane < nd) {
    if (!isnew) {
      my_slot = order[assign];
      Q.det_slot[(size_t)s * H + lane] = my_slot;
    } else if (nt + p < T && p < total_free) {
      const int slot = s2_nth_set_bit(frb, p);
      my_slot = slot;
      const size_t sl = (size_t)s * T + slot;
      Q.used[sl] = 1;
      Q.id[sl] = next_id + p;
      Q.num_obs[sl] = 0;
      Q.has[sl] = 0;
      Q.trk_t_prev[sl] = t;
      Q.height_prev[sl] = -1.0;
      for (int a = 0; a < 3; ++a) Q.root_prev[sl * 3 + a] = 0.0;
      for (int a = 0; a < SE3D_NFU * 9; ++a) Q.vel[sl * SE3D_NFU * 9 + a] = 0.0;
      order[nt + p] = slot;
      Q.det_slot[(size_t)s * H + lane] = slot;
    } else {
      atomicOr(&Q.flags2[s], 1u);  // track capacity exceeded
    }
  }
  if (lane == 0) {
    const int created = min(total_new, min(T - nt, total_free));
    Q.n_tracks[s] = nt + created;
    Q.next_id[s] = next_id + total_new;
  }
  if (Q.direct_out) {
    // Which detections will be published is known here: the LM raises the observation count of the detection's track by one and publishes
    // above g_min_num_obs_track (PP:845-848).  Number them in detection order (= the order s2_track_update_wave publishes in), so that the
    // LM can write each published person once, in place, instead of per detection with a second copy at publishing time (6.7 KB of the
    // 10 KB a published person used to move).
    const bool pubd = my_slot >= 0 && Q.num_obs[(size_t)s * T + my_slot] + 1 > 10;
    const unsigned long long pm_ = __ballot(pubd);
    if (lane < nd) Q.pub[(size_t)s * H + lane] = pubd ? (uint8_t)(1 + __popcll(pm_ & ((1ull << lane) - 1ull))) : 0;
  }
}

// row L + publishing (PP:845-852, 867-903, 909-910) by one wavefront
__device__ __forceinline__ void s2_track_update_wave(const Se3dStage2 &Q, int s, int lane) {
  const int T = Q.T, H = Q.H;
  const int nd = min(Q.n_det[s], H);
  if (nd == 0) return;
  const double t = Q.t[s];
  const bool pb = (lane < nd) && Q.pub[(size_t)s * H + lane];
  const unsigned long long pbm = __ballot(pb);
  const int n_out = __popcll(pbm);
  static_assert(sizeof(se3d_person3d) % 8 == 0 && sizeof(se3d_person3d) / 8 <= 4 * 64, "person copy below: four 8-byte words per lane");
  constexpr int words = (int)(sizeof(se3d_person3d) / 8);
  int p = 0;
  if (!Q.direct_out)   // (fused walk: the LM has written the published persons in place)
  for (unsigned long long m = pbm; m; m &= m - 1ull, ++p) {  // publish in detection order
    const int srcd = __ffsll((long long)m) - 1;
    const size_t src = (size_t)s * H + srcd, dst = (size_t)s * H + p;
    const uint2 *__restrict__ f0 = reinterpret_cast<const uint2 *>(&Q.fused_det[src]), *__restrict__ p0 = reinterpret_cast<const uint2 *>(&Q.pred_det[src]);
    uint2 *__restrict__ f1 = reinterpret_cast<uint2 *>(&Q.fused[dst]), *__restrict__ p1 = reinterpret_cast<uint2 *>(&Q.pred[dst]);
    uint2 a[4], b[4];   // all loads of a person in flight before the first store (one memory round trip per person, not one per word)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int w = lane + 64 * q;
      if (w < words) { a[q] = f0[w]; b[q] = p0[w]; }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int w = lane + 64 * q;
      if (w < words) { f1[w] = a[q]; p1[w] = b[q]; }
    }
    if (lane == 0) { Q.iters[dst] = Q.iters_det[src]; Q.obj[dst] = Q.obj_det[src]; }
  }
  s2_remove_old_wave(Q, s, t, lane);
  s2_wave_mem_sync();  // order[] / n_tracks / fused[].id written above by other lanes of this wavefront
  // merge overlapping tracks (PP:870-903): sequential in i, all later j in parallel (track i is not modified by a merge)
  const int nt = Q.n_tracks[s];
  int *order = Q.order + (size_t)s * T;
  const int slotv = (lane < nt) ? order[lane] : 0;
  unsigned long long alive = nt >= 64 ? ~0ull : ((1ull << nt) - 1ull);
  // lane i holds the merge-distance row of track i (T <= 64: one word): one memory round trip instead of one per track in the loop below
  const unsigned long long my_near = (lane < nt) ? Q.near[((size_t)s * T + slotv) * 4] : 0ull;
  for (int i = 0; i < nt; ++i) {
    if (!((alive >> i) & 1ull)) continue;
    const int slot_i = __builtin_amdgcn_readlane(slotv, i);
    const size_t si = (size_t)s * T + slot_i;
    const unsigned long long row_i = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(my_near >> 32), i) << 32) |
                                     (unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(my_near & 0xFFFFFFFFull), i);
    bool kill = false;
    if (lane > i && lane < nt && ((alive >> lane) & 1ull)) {
      const size_t sj = (size_t)s * T + slotv;
      if ((row_i >> (slotv & 63)) & 1ull) {  // se3d_k_track_near: closer than g_merge_dist_thresh
        kill = true;
        Q.used[sj] = 0;
        const uint32_t idj = (uint32_t)Q.id[sj], idi = (uint32_t)Q.id[si];
        for (int pp2 = 0; pp2 < n_out; ++pp2)  // re-assign id in the persons messages (PP:893-898)
          if (Q.fused[(size_t)s * H + pp2].id == idj) { Q.fused[(size_t)s * H + pp2].id = idi; Q.pred[(size_t)s * H + pp2].id = idi; }
      }
    }
    alive &= ~__ballot(kill);
  }
  const bool kp = (lane < nt) && ((alive >> lane) & 1ull);
  const unsigned long long km = __ballot(kp);
  if (kp) order[__popcll(km & ((1ull << lane) - 1ull))] = slotv;
  if (lane == 0) {
    Q.n_tracks[s] = __popcll(km);
    Q.n_out[s] = n_out;
    Q.t_prev[s] = t;
    Q.frame_nr[s] += 1;
  }
}

// De-normalised previous pose of every track slot of the sequence, prev * height_prev + root_prev (what calc_normed_dist PP:84-101
// and calc_3d_dist PP:103-119 compare against), staged once per phase in LDS: [T][21][3] doubles.
__device__ __forceinline__ void s2_fill_track_pos(const Se3dStage2 &Q, int s, int tid, int nthr, double *tpos) {
  const int T = Q.T;
  for (int idx = tid; idx < T * SE3D_NFU * 3; idx += nthr) {
    const int slot = idx / (SE3D_NFU * 3), r = idx - slot * (SE3D_NFU * 3), a = r % 3;
    const size_t sl = (size_t)s * T + slot;
    tpos[idx] = Q.used[sl] ? Q.prev[sl * SE3D_NFU * 3 + r] * Q.height_prev[sl] + Q.root_prev[sl * 3 + a] : 0.0;
  }
}
// One-wavefront forms over the ACTIVE tracks only (order[0 .. n_tracks)): the position table and the merge-distance matrix of the fused walk.
// s2_fill_track_pos / s2_track_near_flat sweep all T slots (16 rounds of 64 entries / T x T pairs in windows of 64) although a sequence
// has ~6 live tracks; every consumer (track cost, merge distances, the merge loop of s2_track_update_wave) only looks at slots in order[].
__device__ __forceinline__ void s2_fill_track_pos_wave(const Se3dStage2 &Q, int s, int lane, double *tpos) {
  const int T = Q.T;
  const int nt = Q.n_tracks[s];
  const int my_slot = (lane < nt) ? Q.order[(size_t)s * T + lane] : 0;
  for (int base = 0; base < nt * SE3D_NFU * 3; base += 64) {   // uniform trip count: the shuffle below needs its source lanes active
    const int idx = base + lane;
    const int a_ = min(idx / (SE3D_NFU * 3), nt - 1), r = idx - a_ * (SE3D_NFU * 3), a = r % 3;
    const int slot = __shfl(my_slot, a_);
    if (idx < nt * SE3D_NFU * 3) {
      const size_t sl = (size_t)s * T + slot;
      tpos[(size_t)slot * SE3D_NFU * 3 + r] = Q.used[sl] ? Q.prev[sl * SE3D_NFU * 3 + r] * Q.height_prev[sl] + Q.root_prev[sl * 3 + a] : 0.0;
    }
  }
}
__device__ __forceinline__ void s2_track_near_wave(const Se3dStage2 &Q, int s, int lane, const double *tpos) {
  const int T = Q.T;   // <= 64: a row of the bit matrix is one word
  const size_t s0 = (size_t)s * T;
  const int nt = Q.n_tracks[s];
  const int my_slot = (lane < nt) ? Q.order[s0 + lane] : 0;
  unsigned long long cm = 0ull;   // lane a: hits of track a against track b (compact indices)
  for (int p0 = 0; p0 < nt * nt; p0 += 64) {
    const int idx = p0 + lane;
    bool nr = false;
    const int a_ = min(idx / nt, nt - 1), b_ = idx - (idx / nt) * nt;
    const int si = __shfl(my_slot, a_), j = __shfl(my_slot, b_);   // outside the bounds test: the source lanes must be active
    if (idx < nt * nt) {
      if (a_ != b_ && Q.used[s0 + si] && Q.used[s0 + j]) {
        const uint32_t both = Q.has[s0 + si] & Q.has[s0 + j];
        const double *pi = tpos + (size_t)si * SE3D_NFU * 3, *pj = tpos + (size_t)j * SE3D_NFU * 3;
        int nj = 0;
        double dist = 0.0;
        for (int k = 0; k < SE3D_NFU; ++k)
          if ((both >> k) & 1u) {
            const double dx = pi[3 * k] - pj[3 * k], dy = pi[3 * k + 1] - pj[3 * k + 1], dz = pi[3 * k + 2] - pj[3 * k + 2];
            dist += sqrt(dx * dx + dy * dy + dz * dz);
            ++nj;
          }
        nr = (nj > 0 ? dist / nj : 1e6) < 0.20;
      }
    }
    const unsigned long long b = __ballot(nr);   // bit l <-> pair p0 + l
    const int lo = max(lane * nt, p0), hi = min(lane * nt + nt, p0 + 64);  // this lane's row within the window
    if (lane < nt && lo < hi) {
      const int w = hi - lo;
      const unsigned long long m = w >= 64 ? ~0ull : ((1ull << w) - 1ull);
      cm |= ((b >> (lo - p0)) & m) << (lo - lane * nt);
    }
  }
  unsigned long long rowbits = 0ull;   // compact column index -> slot bit
  for (int b_ = 0; b_ < nt; ++b_) {
    const int slot_b = __builtin_amdgcn_readlane(my_slot, b_);
    rowbits |= ((cm >> b_) & 1ull) << slot_b;
  }
  if (lane < nt) {
    unsigned long long *row = Q.near + (s0 + my_slot) * 4;
    row[0] = rowbits; row[1] = 0ull; row[2] = 0ull; row[3] = 0ull;
  }
}
__device__ __forceinline__ void s2_track_cost_lds(const Se3dStage2 &Q, int s, int tid, int nthr, const double *tpos) {
  const int T = Q.T, H = Q.H;
  const int nd = min(Q.n_det[s], H), nt = Q.n_tracks[s];
  const se3d_person3d *dets = Q.dets + (size_t)s * H;
  double *C = Q.C + (size_t)s * H * T;
  const double t = Q.t[s];
  for (int idx = tid; idx < nd * nt; idx += nthr) {
    const int det = idx % nd, tk = idx / nd;
    const int slot = Q.order[(size_t)s * T + tk];
    const size_t sl = (size_t)s * T + slot;
    const uint32_t has = Q.has[sl];
    const double delta_t = t - Q.trk_t_prev[sl];
    const double *pp = tpos + (size_t)slot * SE3D_NFU * 3;
    int nj = 0;
    double dist = 0.0;
    constexpr int CB = 21;   // keypoints in flight per memory round trip (7 when the walk was compiled for 168 VGPRs)
    static_assert(SE3D_NFU % CB == 0, "joints are fetched CB at a time");
    for (int k0 = 0; k0 < SE3D_NFU; k0 += CB) {  // (a load per joint behind its own test: 21 trips)
      double jx[CB], jy[CB], jz[CB];
      float sc[CB];
#pragma unroll
      for (int q = 0; q < CB; ++q) {
        const se3d_keypoint3d &kp = dets[det].keypoints[k0 + q];
        jx[q] = kp.joint[0]; jy[q] = kp.joint[1]; jz[q] = kp.joint[2]; sc[q] = kp.score;
      }
#pragma unroll
      for (int q = 0; q < CB; ++q) {
        const int k = k0 + q;
        if (sc[q] > 0.10f && ((has >> k) & 1u)) {
          const double dx = jx[q] - pp[3 * k], dy = jy[q] - pp[3 * k + 1], dz = jz[q] - pp[3 * k + 2];
          dist += sqrt(dx * dx + dy * dy + dz * dz) / (c_vel_sigmas[k] * delta_t);
          ++nj;
        }
      }
    }
    C[det + (size_t)nd * tk] = nj > 0 ? dist / nj : 1e6;
  }
}
// row `si` of the merge-distance bit matrix (T <= 64: word 0 only) from the staged positions; one wavefront, lane <-> slot j
__device__ __forceinline__ void s2_track_near_lds(const Se3dStage2 &Q, int si, int s, int lane, const double *tpos) {
  const int T = Q.T;
  const size_t sli = (size_t)s * T + si;
  unsigned long long *row = Q.near + sli * 4;
  if (!Q.used[sli]) { if (lane < 4) row[lane] = 0ull; return; }
  const uint32_t hasi = Q.has[sli];
  const double *pi = tpos + (size_t)si * SE3D_NFU * 3;
  bool nr = false;
  if (lane < T && lane != si && Q.used[(size_t)s * T + lane]) {
    const uint32_t both = hasi & Q.has[(size_t)s * T + lane];
    const double *pj = tpos + (size_t)lane * SE3D_NFU * 3;
    int nj = 0;
    double dist = 0.0;
    for (int k = 0; k < SE3D_NFU; ++k)
      if ((both >> k) & 1u) {
        const double dx = pi[3 * k] - pj[3 * k], dy = pi[3 * k + 1] - pj[3 * k + 1], dz = pi[3 * k + 2] - pj[3 * k + 2];
        dist += sqrt(dx * dx + dy * dy + dz * dz);
        ++nj;
      }
    nr = (nj > 0 ? dist / nj : 1e6) < 0.20;
  }
  const unsigned long long b = __ballot(nr);
  if (lane == 0) row[0] = b;
  if (lane >= 1 && lane < 4) row[lane] = 0ull;
}

// the whole merge-distance bit matrix by one wavefront (T <= 64): (si, j) pairs flattened over the lanes, row bits collected from the
// ballots of each pass by the lane that owns the row.  Same decisions as s2_track_near_lds row by row.
__device__ __forceinline__ void s2_track_near_flat(const Se3dStage2 &Q, int s, int lane, const double *tpos) {
  const int T = Q.T;
  const size_t s0 = (size_t)s * T;
  unsigned long long rowbits = 0ull;
  for (int p0 = 0; p0 < T * T; p0 += 64) {
    const int idx = p0 + lane;
    bool nr = false;
    if (idx < T * T) {
      const int si = idx / T, j = idx - si * T;
      if (si != j && Q.used[s0 + si] && Q.used[s0 + j]) {
        const uint32_t both = Q.has[s0 + si] & Q.has[s0 + j];
        const double *pi = tpos + (size_t)si * SE3D_NFU * 3, *pj = tpos + (size_t)j * SE3D_NFU * 3;
        int nj = 0;
        double dist = 0.0;
        for (int k = 0; k < SE3D_NFU; ++k)
          if ((both >> k) & 1u) {
            const double dx = pi[3 * k] - pj[3 * k], dy = pi[3 * k + 1] - pj[3 * k + 1], dz = pi[3 * k + 2] - pj[3 * k + 2];
            dist += sqrt(dx * dx + dy * dy + dz * dz);
            ++nj;
          }
        nr = (nj > 0 ? dist / nj : 1e6) < 0.20;
      }
    }
    const unsigned long long b = __ballot(nr);   // bit l <-> pair p0 + l
    const int lo = max(lane * T, p0), hi = min(lane * T + T, p0 + 64);  // this lane's row within the window
    if (lane < T && lo < hi) {
      const int w = hi - lo;
      const unsigned long long m = w >= 64 ? ~0ull : ((1ull << w) - 1ull);
      rowbits |= ((b >> (lo - p0)) & m) << (lo - lane * T);
    }
  }
  if (lane < T) {
    unsigned long long *row = Q.near + (s0 + lane) * 4;
    row[0] = rowbits; row[1] = 0ull; row[2] = 0ull; row[3] = 0ull;
  }
}

#ifndef S2_TRIO_WAVES
#define S2_TRIO_WAVES 2   // wavefronts per SIMD the trio walk is compiled for (256 VGPRs)
#endif
#ifndef S2_WALK_WAVES
#define S2_WALK_WAVES 4
#endif
#ifdef SE3D_LM_STAMPS
#define WK_STAMP_INIT unsigned long long wk_prev_ = __builtin_readcyclecounter();
#define WK_STAMP(idx) { const unsigned long long now_ = __builtin_readcyclecounter(); if (threadIdx.x == 0 && (blockIdx.x & 63) == 0) atomicAdd(&g_lm_stamps[idx], now_ - wk_prev_); wk_prev_ = now_; }
#else
#define WK_STAMP_INIT
#define WK_STAMP(idx)
#endif
// S2_WALK_SEQS sequences share a workgroup: their persons feed one LM work queue (the slowest person of a sequence-step no longer
// idles three wavefronts: the other sequence's persons fill them), their association / lifecycle phases run on different
// wavefronts at the same time, and a sequence costs half the registers / wave slots of a one-sequence workgroup.
#ifndef S2_WALK_SEQS
#define S2_WALK_SEQS 2
#endif
// TRIO: three persons per LM wavefront (s2_lm_trio_body); otherwise one person per wavefront (s2_lm_body<3>, kept as A/B evidence)
// WV wavefronts per workgroup, SQ sequences per workgroup.  <4, 2>: few sequences per launch (latency: a sequence-step spreads its
// persons over four wavefronts).  <1, 1>: one wavefront walks one sequence on its own — no workgroup barriers, no wavefront ever
// waits for a slower one (LM trial counts vary from 2 to 60), every resident wavefront slot does useful work: the throughput shape
// for launches of many sequences.  Its LM workspace, Munkres scratch and track-position table share one LDS region (the phases of
// a single wavefront are sequential).
template <bool TRIO, int WV, int SQ>
__global__ __launch_bounds__(64 * WV) __attribute__((amdgpu_waves_per_eu(TRIO ? S2_TRIO_WAVES : 3, TRIO ? S2_TRIO_WAVES : 3))) void se3d_k_stage2_walk(Se3dStage2 Q, Se3dWalk Wk, int nseq) {
  using WaveLds = typename std::conditional<TRIO, LmTrioLds, LmTreeLds>::type;
  extern __shared__ __align__(16) unsigned char s2_walk_smem[];
  __shared__ int s_next_person;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sb = blockIdx.x * SQ, ns = min(SQ, nseq - sb);  // sequences sb .. sb + ns - 1 (local index 0 .. ns - 1)
  const int H = Q.H, T = Q.T;
  constexpr int NT = 64 * WV;
  // dynamic LDS: [WV] LM workspaces | [SEQS][T][63] de-normalised track positions | optional per-track state [SEQS][T]...
  const size_t tpos_stride = (size_t)T * SE3D_NFU * 3;
  // one wavefront: one region for everything (same formula as se3d_launch_stage2_walk)
  const size_t shared_bytes = (max(max(sizeof(WaveLds), tpos_stride * sizeof(double)), (size_t)H * T * sizeof(double)) + 15) & ~(size_t)15;
  double *tpos = reinterpret_cast<double *>(s2_walk_smem + (WV == 1 ? 0 : WV * sizeof(WaveLds)));
  // Per-track state of the sequences lives in LDS for the duration of the walk when it fits (Wk.lds_state): every phase touches it
  // through the same Se3dStage2 pointers (generic pointers), at LDS instead of HBM/L2 latency.
  Se3dStage2 Qs = Q;
  double *l_root = nullptr, *l_height = nullptr, *l_tprev = nullptr, *l_prev = nullptr;
  uint32_t *l_has = nullptr;
  int32_t *l_id = nullptr, *l_nobs = nullptr;
  uint8_t *l_used = nullptr;
  const int TT = SQ * T;       // LDS state capacity (track slots of all sequences of the workgroup)
  const int tn = ns * T;                 // track slots in use
  const size_t g0 = (size_t)sb * T;
  if (Wk.lds_state) {
    double *b = WV == 1 ? reinterpret_cast<double *>(s2_walk_smem + shared_bytes) : tpos + SQ * tpos_stride;
    l_root = b; b += (size_t)TT * 3;
    l_height = b; b += TT;
    l_tprev = b; b += TT;
    l_has = reinterpret_cast<uint32_t *>(b);
    l_id = reinterpret_cast<int32_t *>(l_has + TT);
    l_nobs = l_id + TT;
    l_used = reinterpret_cast<uint8_t *>(l_nobs + TT);
    if (Wk.lds_state & 2) {  // the normalised previous poses too (T x 63 doubles per sequence), behind the small state (8-byte aligned)
      l_prev = reinterpret_cast<double *>(reinterpret_cast<unsigned char *>(b) + (((size_t)TT * 16 + 7) & ~(size_t)7));
      for (int i = tid; i < tn * SE3D_NFU * 3; i += NT) l_prev[i] = Q.prev[g0 * SE3D_NFU * 3 + i];
    }
    for (int i = tid; i < tn * 3; i += NT) l_root[i] = Q.root_prev[g0 * 3 + i];
    for (int i = tid; i < tn; i += NT) {
      l_height[i] = Q.height_prev[g0 + i]; l_tprev[i] = Q.trk_t_prev[g0 + i]; l_has[i] = Q.has[g0 + i];
      l_id[i] = Q.id[g0 + i]; l_nobs[i] = Q.num_obs[g0 + i]; l_used[i] = Q.used[g0 + i];
    }
    Qs.root_prev = l_root; Qs.height_prev = l_height; Qs.trk_t_prev = l_tprev;
    Qs.has = l_has; Qs.id = l_id; Qs.num_obs = l_nobs; Qs.used = l_used;
    __syncthreads();
  } else {
    Qs.root_prev += g0 * 3; Qs.height_prev += g0; Qs.trk_t_prev += g0;
    Qs.has += g0; Qs.id += g0; Qs.num_obs += g0; Qs.used += g0;
  }
  // (lds_state & 4, one-wavefront shape) the per-step bookkeeping too: cost matrix, merge-distance rows, track order, detection -> slot /
  // publish tables and the per-sequence scalars.  In global memory each of them is a store -> fence -> load round trip to L2 between two
  // phases of the same wavefront, a dozen per time step.
  double *b_C = nullptr, *b_tprev = nullptr, *b_fb = nullptr;
  unsigned long long *b_near = nullptr;
  int32_t *b_order = nullptr, *b_detslot = nullptr, *b_ntr = nullptr, *b_nid = nullptr, *b_fnr = nullptr;
  uint8_t *b_pub = nullptr;
  if (Wk.lds_state & 4) {
    unsigned char *base = s2_walk_smem + Wk.book_offset;
    b_C = reinterpret_cast<double *>(base); base += (size_t)SQ * H * T * sizeof(double);
    b_near = reinterpret_cast<unsigned long long *>(base); base += (size_t)TT * 4 * sizeof(unsigned long long);
    b_tprev = reinterpret_cast<double *>(base); base += (size_t)SQ * sizeof(double);
    b_fb = reinterpret_cast<double *>(base); base += (size_t)SQ * 3 * sizeof(double);
    b_order = reinterpret_cast<int32_t *>(base); base += (size_t)TT * 4;
    b_detslot = reinterpret_cast<int32_t *>(base); base += (size_t)SQ * H * 4;
    b_ntr = reinterpret_cast<int32_t *>(base); base += (size_t)SQ * 4;
    b_nid = reinterpret_cast<int32_t *>(base); base += (size_t)SQ * 4;
    b_fnr = reinterpret_cast<int32_t *>(base); base += (size_t)SQ * 4;
    b_pub = reinterpret_cast<uint8_t *>(base);
    for (int i = tid; i < tn; i += NT) b_order[i] = Q.order[g0 + i];
    for (int i = tid; i < ns; i += NT) {
      b_ntr[i] = Q.n_tracks[sb + i]; b_nid[i] = Q.next_id[sb + i]; b_fnr[i] = Q.frame_nr[sb + i]; b_tprev[i] = Q.t_prev[sb + i];
      for (int a = 0; a < 3; ++a) b_fb[i * 3 + a] = Q.fb_buf[(size_t)(sb + i) * 3 + a];
    }
    __syncthreads();
  }
  // every per-sequence array is rebased to the first sequence of the workgroup: the phase bodies run with the local sequence index
  // the normalised previous pose (0.5 KB per track) and the velocity ring (1.5 KB), touched once per person and step, stay in HBM / L2;
  // the phases that compare against every track (cost, merge distance) read the de-normalised table `tpos` in LDS
  if (l_prev) Qs.prev = l_prev; else Qs.prev += g0 * SE3D_NFU * 3;
  Qs.vel += g0 * SE3D_NFU * 9;
  if (Wk.lds_state & 4) {
    Qs.fb_buf = b_fb; Qs.t_prev = b_tprev; Qs.next_id = b_nid; Qs.frame_nr = b_fnr; Qs.n_tracks = b_ntr; Qs.order = b_order;
    Qs.C = b_C; Qs.near = b_near; Qs.det_slot = b_detslot; Qs.pub = b_pub;
  } else {
    Qs.fb_buf += (size_t)sb * 3; Qs.t_prev += sb; Qs.next_id += sb; Qs.frame_nr += sb; Qs.n_tracks += sb; Qs.order += g0;
    Qs.C += (size_t)sb * H * T; Qs.near += g0 * 4; Qs.det_slot += (size_t)sb * H; Qs.pub += (size_t)sb * H;
  }
  Qs.W += (size_t)sb * H * T;
  Qs.fused_det += (size_t)sb * H; Qs.pred_det += (size_t)sb * H; Qs.iters_det += (size_t)sb * H; Qs.obj_det += (size_t)sb * H; Qs.flags2 += sb;
  const bool par_assoc = WV > 1 && (size_t)H * T * sizeof(double) <= sizeof(WaveLds);  // each association wavefront has its own Munkres scratch
  for (int k = 0; k < Wk.n_steps; ++k) {
    const size_t f0 = (size_t)Wk.slot0[k] + sb;
    Se3dStage2 q = Qs;
    q.dets = Wk.outP + f0 * H;
    q.n_det = Wk.out_count + f0;
    q.fb_delay = Wk.fb + f0 * Q.N;
    q.t = Wk.t + (size_t)k * Wk.t_stride + sb;
    q.fused = Wk.Wf + f0 * H;
    q.pred = Wk.Wp + f0 * H;
    q.iters = Wk.Wit + f0 * H;
    q.obj = Wk.Wobj + f0 * H;
    q.n_out = Wk.Wn + f0;
    q.pred_dt = Wk.Wdt + f0;
    q.direct_out = 1;
    int nd[SQ];
#pragma unroll
    for (int s = 0; s < SQ; ++s) nd[s] = (s < ns) ? min(q.n_det[s], H) : 0;
    WK_STAMP_INIT
    if constexpr (WV == 1) {
      if (nd[0] > 0) s2_fill_track_pos_wave(q, 0, lane, tpos);
    } else {
#pragma unroll
      for (int s = 0; s < SQ; ++s)
        if (nd[s] > 0) s2_fill_track_pos(q, s, tid, NT, tpos + s * tpos_stride);
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < SQ; ++s)
      if (nd[s] > 0) s2_track_cost_lds(q, s, tid, NT, tpos + s * tpos_stride);
    __syncthreads();
    WK_STAMP(10)  // track cost
    if (par_assoc) {
      if (wave < ns) s2_track_assoc_wave(q, wave, lane, reinterpret_cast<double *>(s2_walk_smem + wave * sizeof(WaveLds)));
    } else if (wave == 0) {
      for (int s = 0; s < ns; ++s) s2_track_assoc_wave(q, s, lane, reinterpret_cast<double *>(s2_walk_smem));
    }
    if (tid == 0) s_next_person = 0;
    __syncthreads();
    if constexpr (SQ == 1) {   // the step's per-sequence scalars once, not per LM pass
      q.u_pred_dt = q.pred_dt[0];
      q.u_fslot = q.frame_nr[0] % 3;
      q.u_t_prev = q.t_prev[0];
      q.u_valid = 1;
    }
    WK_STAMP(11)  // association (Munkres)
    {
      int total = 0;
#pragma unroll
      for (int s = 0; s < SQ; ++s) total += nd[s];
      if constexpr (TRIO) {
        LmTrioLds &W = reinterpret_cast<LmTrioLds *>(s2_walk_smem)[wave];
        const int chunk = min(3, max(1, (total + WV - 1) / WV));  // persons per grab: spread them over the wavefronts
        for (;;) {
          int idx = 0;
          if (lane == 0) idx = atomicAdd(&s_next_person, chunk);
          idx = __builtin_amdgcn_readfirstlane(idx);
          if (idx >= total) break;
          const int g3 = lane / SE3D_NFU;
          const bool have = g3 < chunk && idx + g3 < total;
          int s = 0, i = idx + g3;
#pragma unroll
          for (int ss = 0; ss + 1 < SQ; ++ss)
            if (s == ss && i >= nd[ss]) { i -= nd[ss]; s = ss + 1; }
          s2_lm_trio_body(q, s, i, have, lane, W);
          __builtin_amdgcn_wave_barrier();
        }
      } else {
        LmTreeLds &L = reinterpret_cast<LmTreeLds *>(s2_walk_smem)[wave];
        if (lane == 0) {  // the association phase used this LDS as Munkres scratch
          L.bind(Q.lm_scratch + ((size_t)blockIdx.x * WV + wave) * S2_SCHUR_SCRATCH_DOUBLES);
        }
        __builtin_amdgcn_wave_barrier();
        for (;;) {  // persons of all sequences are handed out dynamically: LM iteration counts vary from 2 to 60
          int idx = 0;
          if (lane == 0) idx = atomicAdd(&s_next_person, 1);
          idx = __builtin_amdgcn_readfirstlane(idx);
          if (idx >= total) break;
          int s = 0, i = idx;
#pragma unroll
          for (int ss = 0; ss + 1 < SQ; ++ss)
            if (s == ss && i >= nd[ss]) { i -= nd[ss]; s = ss + 1; }
          s2_lm_body<3, LmTreeLds>(q, i, s, lane, L);
          __builtin_amdgcn_wave_barrier();
        }
      }
    }
    __syncthreads();
    WK_STAMP(12)  // LM
    if constexpr (WV == 1) {
      if (nd[0] > 0) s2_fill_track_pos_wave(q, 0, lane, tpos);  // the LM phase has updated the tracks
    } else {
#pragma unroll
      for (int s = 0; s < SQ; ++s)
        if (nd[s] > 0) s2_fill_track_pos(q, s, tid, NT, tpos + s * tpos_stride);  // the LM phase has updated the tracks
    }
    __syncthreads();
    if constexpr (WV == 1) {
      if (nd[0] > 0) s2_track_near_wave(q, 0, lane, tpos);
    } else
    for (int it = wave; it < ns * T; it += WV) {
      const int s = it / T, si = it - s * T;
      int nds = 0;
#pragma unroll
      for (int ss = 0; ss < SQ; ++ss) nds = (s == ss) ? nd[ss] : nds;
      if (nds > 0) s2_track_near_lds(q, si, s, lane, tpos + s * tpos_stride);
    }
    __syncthreads();
    WK_STAMP(13)  // merge-distance bit matrix
    if (wave < ns) {
      int ndw = 0;
#pragma unroll
      for (int ss = 0; ss < SQ; ++ss) ndw = (wave == ss) ? nd[ss] : ndw;
      if (ndw > 0) s2_track_update_wave(q, wave, lane);
    }
    __syncthreads();
    WK_STAMP(14)  // publish + lifecycle
  }
  if (Wk.lds_state & 4) {  // persistent bookkeeping back to global memory (C, near, det_slot, pub are per-step scratch)
    __syncthreads();
    for (int i = tid; i < tn; i += NT) Q.order[g0 + i] = b_order[i];
    for (int i = tid; i < ns; i += NT) {
      Q.n_tracks[sb + i] = b_ntr[i]; Q.next_id[sb + i] = b_nid[i]; Q.frame_nr[sb + i] = b_fnr[i]; Q.t_prev[sb + i] = b_tprev[i];
      for (int a = 0; a < 3; ++a) Q.fb_buf[(size_t)(sb + i) * 3 + a] = b_fb[i * 3 + a];
    }
  }
  if (Wk.lds_state) {  // write the track state back
    if (l_prev)
      for (int i = tid; i < tn * SE3D_NFU * 3; i += NT) Q.prev[g0 * SE3D_NFU * 3 + i] = l_prev[i];
    for (int i = tid; i < tn * 3; i += NT) Q.root_prev[g0 * 3 + i] = l_root[i];
    for (int i = tid; i < tn; i += NT) {
      Q.height_prev[g0 + i] = l_height[i]; Q.trk_t_prev[g0 + i] = l_tprev[i]; Q.has[g0 + i] = l_has[i];
      Q.id[g0 + i] = l_id[i]; Q.num_obs[g0 + i] = l_nobs[i]; Q.used[g0 + i] = l_used[i];
    }
  }
}
static_assert(S2_SCHUR_SCRATCH_DOUBLES == 1865, "se3d_api.hip allocates lm_scratch with this size");
static_assert(S2_WALK_SEQS >= 1 && S2_WALK_SEQS <= S2_WALK_WAVES, "one association / lifecycle wavefront per sequence of the workgroup");
#define S2_WALK_LDS_MAX (160 * 1024 - 4096)  // 160 KB per CU minus the kernel's static LDS
template <bool TRIO, int WV, int SQ>
static bool s2_launch_walk_shape(const Se3dStage2 &q, const Se3dWalk &Wk, int nseq, hipStream_t st, bool book_in_lds) {
  const int lds_max = S2_WALK_LDS_MAX;  // raised per device by se3d_stage2_init_device
  using WaveLds = typename std::conditional<TRIO, LmTrioLds, LmTreeLds>::type;
  const size_t tpos_bytes = (size_t)SQ * q.T * SE3D_NFU * 3 * sizeof(double);
  const size_t work_bytes = WV == 1 ? ((std::max(std::max(sizeof(WaveLds), tpos_bytes), (size_t)q.H * q.T * sizeof(double)) + 15) & ~(size_t)15)
                                    : WV * sizeof(WaveLds) + tpos_bytes;
  // small per-track state in LDS when it fits: root 3 + height + t_prev doubles, 4 words
  const size_t state_bytes = (size_t)SQ * q.T * (5 * sizeof(double) + 16);
  Se3dWalk w = Wk;
  w.lds_state = (work_bytes + state_bytes <= (size_t)lds_max) ? 1 : 0;
  // ... and the normalised previous poses (0.5 KB per track) when the one-wavefront shape still fits eight workgroups per CU with them
  const size_t prev_bytes = (size_t)SQ * q.T * SE3D_NFU * 3 * sizeof(double) + 8;
  if (WV == 1 && w.lds_state && work_bytes + state_bytes + prev_bytes <= 19 * 1024) w.lds_state |= 2;
  size_t dyn = work_bytes + (w.lds_state ? state_bytes : 0) + ((w.lds_state & 2) ? prev_bytes : 0);
  // ... and the per-step bookkeeping (cost matrix, merge rows, order, slot / publish tables, scalars) while eight workgroups still fit a CU
  const size_t book_bytes = (size_t)SQ * q.H * q.T * 8 + (size_t)SQ * q.T * (32 + 4) + (size_t)SQ * (8 + 24 + 12) + (size_t)SQ * q.H * 5 + 32;
  dyn = (dyn + 15) & ~(size_t)15;
  if (WV == 1 && (w.lds_state & 2) && dyn + book_bytes <= 20 * 1024 && book_in_lds) {   // (options.walk_book = 1 keeps the bookkeeping in global memory: parity tests)
    w.lds_state |= 4;
    w.book_offset = (int)dyn;
    dyn += book_bytes;
  }
  if (dyn > (size_t)lds_max) return false;
  hipLaunchKernelGGL((se3d_k_stage2_walk<TRIO, WV, SQ>), dim3((nseq + SQ - 1) / SQ), dim3(64 * WV), dyn, st, q, w, nseq);
  return true;
}

#ifdef SE3D_LM_STAMPS
extern "C" void se3d_diag_lm_ablate(int bits) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lm_ablate), &bits, sizeof(int)); }
extern "C" void se3d_diag_lm_stamps(unsigned long long *out16, int reset) {
  (void)hipDeviceSynchronize();
  (void)hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_lm_stamps), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lm_stamps), z, sizeof(z)); }
}
#endif
// kernel template argument of the per-step LM for options.lm_form: 3 = forest elimination (default), 2 = forest solve + MFMA marginals, 0 = dense J^T J on MFMA
static int s2_lm_mode(const se3d_options &o) { return o.lm_form == SE3D_LM_DENSE ? 0 : o.lm_form == SE3D_LM_TREE_MFMA ? 2 : 3; }

// true: the fused kernel was launched; false: not applicable (caller falls back to per-step launches)
extern "C" bool se3d_launch_stage2_walk(const Se3dStage2 &Q, const se3d_options &o, const Se3dWalk &Wk, int nseq, hipStream_t st) {
  const bool trio = o.walk_lm != SE3D_WALK_LM_SOLO;   // solo: one person per LM wavefront (the per-step device code)
  const bool ok = o.walk_form != SE3D_WALK_STEPS && s2_lm_mode(o) == 3;
  // The fused walk is right for many short sequences of few persons.  Large association problems (> 64) and few sequences of many
  // persons keep the per-step kernels (a wavefront per person over the whole GPU).
  if (!ok || Q.H > 64 || Q.T > 64 || nseq < 1 || (Q.H > 16 && nseq < 32)) return false;
  Se3dStage2 q = Q;
  q.seq0 = 0;
  // one wavefront per sequence once a launch brings enough sequences to fill wavefront slots that way (and a sequence-step is a few
  // trios at most); otherwise the latency shape (options.walk_shape forces one of them)
  bool wave_shape = trio && Q.H <= 16 && nseq >= 256;
  if (o.walk_shape) wave_shape = trio && o.walk_shape == SE3D_WALK_WAVE;
  const bool book = o.walk_book == 0;
  if (wave_shape && s2_launch_walk_shape<true, 1, 1>(q, Wk, nseq, st, book)) return true;
  return trio ? s2_launch_walk_shape<true, S2_WALK_WAVES, S2_WALK_SEQS>(q, Wk, nseq, st, book)
              : s2_launch_walk_shape<false, S2_WALK_WAVES, S2_WALK_SEQS>(q, Wk, nseq, st, book);
}

// Diagnostic: the fusion-joint indices the kernels use by name (FU_*) and the device's vel_sigmas table, for the test that compares them
// with the reference's fusion_body_parts.h (tests/golden/fusion_body_parts.json).  -1: an index the kernels never refer to by name.
extern "C" hipError_t se3d_read_fusion_constants(int32_t *indices22, double *vel_sigmas21) {
  for (int i = 0; i < 22; ++i) indices22[i] = -1;
  indices22[0] = FU_Nose; indices22[1] = FU_Neck; indices22[2] = FU_RShoulder; indices22[5] = FU_LShoulder; indices22[8] = FU_MidHip;
  indices22[9] = FU_RHip; indices22[11] = FU_RAnkle; indices22[12] = FU_LHip; indices22[14] = FU_LAnkle; indices22[20] = FU_Belly; indices22[21] = SE3D_NFU;
  return hipMemcpyFromSymbol(vel_sigmas21, HIP_SYMBOL(c_vel_sigmas), sizeof(double) * SE3D_NFU);
}

// Kernel attributes are per DEVICE in HIP: called by se3d_create for the context's device (after hipSetDevice), not once per process.
extern "C" hipError_t se3d_stage2_init_device(void) {
  const struct { const void *fn; int bytes; } attrs[] = {
      {reinterpret_cast<const void *>(se3d_k_track_assoc), 128 * 1024},
      {reinterpret_cast<const void *>(se3d_k_lm_refine<0>), (int)sizeof(LmLds)},
      {reinterpret_cast<const void *>(se3d_k_lm_refine<2>), (int)sizeof(LmSchurLds)},
      {reinterpret_cast<const void *>(se3d_k_lm_refine<3>), (int)sizeof(LmSchurLds)},
      {reinterpret_cast<const void *>(se3d_k_stage2_walk<true, 1, 1>), S2_WALK_LDS_MAX},
      {reinterpret_cast<const void *>(se3d_k_stage2_walk<true, S2_WALK_WAVES, S2_WALK_SEQS>), S2_WALK_LDS_MAX},
      {reinterpret_cast<const void *>(se3d_k_stage2_walk<false, S2_WALK_WAVES, S2_WALK_SEQS>), S2_WALK_LDS_MAX},
  };
  for (const auto &a : attrs) {
    const hipError_t e = hipFuncSetAttribute(a.fn, hipFuncAttributeMaxDynamicSharedMemorySize, a.bytes);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// Hypothesis shard, stage 2: after the (sharded) LM step every rank packs its detections' records (i % world == rank -> slot i / world),
// the ranks all-gather them and every rank applies the other ranks' records: outputs per detection and the state of the updated track.
__global__ __launch_bounds__(64) void se3d_k_shard_pack_lm(Se3dStage2 Q, Se3dShardRec *__restrict__ out) {
  const int j = blockIdx.x, s = blockIdx.y + Q.seq0, Hs = gridDim.x, i = j * Q.shard_world + Q.shard_rank, lane = threadIdx.x;
  Se3dShardRec *rec = out + (size_t)blockIdx.y * Hs + j;
  if (i >= min(Q.n_det[s], Q.H)) return;
  const size_t di = (size_t)s * Q.H + i;
  constexpr int NW = (int)(sizeof(se3d_person3d) / 4);
  const uint32_t *sf = reinterpret_cast<const uint32_t *>(Q.fused_det + di), *sp = reinterpret_cast<const uint32_t *>(Q.pred_det + di);
  uint32_t *df = reinterpret_cast<uint32_t *>(&rec->fused), *dp = reinterpret_cast<uint32_t *>(&rec->pred);
  for (int k = lane; k < NW; k += 64) { df[k] = sf[k]; dp[k] = sp[k]; }
  const int slot = Q.det_slot[di];
  if (lane == 0) { rec->iters = Q.iters_det[di]; rec->obj = Q.obj_det[di]; rec->pub = Q.pub[di]; }
  if (slot < 0) return;
  const size_t sl = (size_t)s * Q.T + slot;
  for (int k = lane; k < SE3D_NFU * 3; k += 64) rec->prev[k] = Q.prev[sl * SE3D_NFU * 3 + k];
  for (int k = lane; k < SE3D_NFU * 9; k += 64) rec->vel[k] = Q.vel[sl * SE3D_NFU * 9 + k];
  if (lane < 3) rec->root[lane] = Q.root_prev[sl * 3 + lane];
  if (lane == 0) { rec->t_prev = Q.trk_t_prev[sl]; rec->height = Q.height_prev[sl]; rec->num_obs = Q.num_obs[sl]; rec->has = Q.has[sl]; }
}
__global__ __launch_bounds__(64) void se3d_k_shard_unpack_lm(Se3dStage2 Q, const Se3dShardRec *__restrict__ gathered, int Hs, int nseq) {
  const int r = blockIdx.x / Hs, j = blockIdx.x - r * Hs, s = blockIdx.y + Q.seq0, i = j * Q.shard_world + r, lane = threadIdx.x;
  if (r == Q.shard_rank || i >= min(Q.n_det[s], Q.H)) return;
  const Se3dShardRec *rec = gathered + ((size_t)r * nseq + blockIdx.y) * Hs + j;
  const size_t di = (size_t)s * Q.H + i;
  constexpr int NW = (int)(sizeof(se3d_person3d) / 4);
  const uint32_t *sf = reinterpret_cast<const uint32_t *>(&rec->fused), *sp = reinterpret_cast<const uint32_t *>(&rec->pred);
  uint32_t *df = reinterpret_cast<uint32_t *>(Q.fused_det + di), *dp = reinterpret_cast<uint32_t *>(Q.pred_det + di);
  for (int k = lane; k < NW; k += 64) { df[k] = sf[k]; dp[k] = sp[k]; }
  const int slot = Q.det_slot[di];
  if (lane == 0) { Q.iters_det[di] = rec->iters; Q.obj_det[di] = rec->obj; Q.pub[di] = rec->pub; }
  if (slot < 0) return;
  const size_t sl = (size_t)s * Q.T + slot;
  for (int k = lane; k < SE3D_NFU * 3; k += 64) Q.prev[sl * SE3D_NFU * 3 + k] = rec->prev[k];
  for (int k = lane; k < SE3D_NFU * 9; k += 64) Q.vel[sl * SE3D_NFU * 9 + k] = rec->vel[k];
  if (lane < 3) Q.root_prev[sl * 3 + lane] = rec->root[lane];
  if (lane == 0) { Q.trk_t_prev[sl] = rec->t_prev; Q.height_prev[sl] = rec->height; Q.num_obs[sl] = rec->num_obs; Q.has[sl] = rec->has; }
}
extern "C" void se3d_launch_shard_pack_lm(const Se3dStage2 &Q, int nseq, Se3dShardRec *out, hipStream_t st) {
  const int Hs = (Q.H + Q.shard_world - 1) / Q.shard_world;
  hipLaunchKernelGGL(se3d_k_shard_pack_lm, dim3(Hs, nseq), dim3(64), 0, st, Q, out);
}
extern "C" void se3d_launch_shard_unpack_lm(const Se3dStage2 &Q, int nseq, const Se3dShardRec *gathered, hipStream_t st) {
  const int Hs = (Q.H + Q.shard_world - 1) / Q.shard_world;
  hipLaunchKernelGGL(se3d_k_shard_unpack_lm, dim3(Hs * Q.shard_world, nseq), dim3(64), 0, st, Q, gathered, Hs, nseq);
}

// phases: 1 = track costs + association + LM refinement, 2 = merge-distance matrix + publish / lifecycle
extern "C" void se3d_launch_stage2(const Se3dStage2 &Q, const se3d_options &o, int seq0, int nseq, hipStream_t st, hipEvent_t *ev, int phases) {
  Se3dStage2 q = Q;
  q.seq0 = seq0;
  const int lm_mode = s2_lm_mode(o);
  if (ev) (void)hipEventRecord(ev[0], st);
  int dynW = (int)((size_t)q.H * q.T * sizeof(double));
  if (dynW > 128 * 1024) dynW = 128 * 1024;
  if (phases & 1) {
    hipLaunchKernelGGL(se3d_k_track_cost, dim3(q.T, nseq), dim3(128), 0, st, q);
    hipLaunchKernelGGL(se3d_k_track_assoc, dim3(nseq), dim3(256), (size_t)dynW, st, q, dynW);
  }
  if (ev) (void)hipEventRecord(ev[1], st);
  if (phases & 1) {
    if (lm_mode == 0) hipLaunchKernelGGL(se3d_k_lm_refine<0>, dim3(q.H, nseq), dim3(64), sizeof(LmLds), st, q);
    else if (lm_mode == 2) hipLaunchKernelGGL(se3d_k_lm_refine<2>, dim3(q.H, nseq), dim3(64), sizeof(LmSchurLds), st, q);
    else hipLaunchKernelGGL(se3d_k_lm_refine<3>, dim3(q.H, nseq), dim3(64), sizeof(LmSchurLds), st, q);
  }
  if (ev) (void)hipEventRecord(ev[2], st);
  if (phases & 2) {
    hipLaunchKernelGGL(se3d_k_track_near, dim3(q.T, nseq), dim3(64), 0, st, q);
    hipLaunchKernelGGL(se3d_k_track_update, dim3(nseq), dim3(256), 0, st, q);
  }
  if (ev) (void)hipEventRecord(ev[3], st);
}
