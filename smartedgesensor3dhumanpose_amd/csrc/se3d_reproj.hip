// se3d_reproj.hip — feedback reprojection (pose_reprojection/src/skeleton_reproj_mult_node.cpp:139-235) for MI355X.
//
// One 32-lane half-wave per (person, camera): lane j < 17 owns model joint j, draws the 7 sigma points of the joint's 3-D
// covariance (REPROJ:62-75), projects them with the camera's double-precision pose and pinhole matrix, and forms the weighted
// mean / 2x2 covariance (REPROJ:178-205).  Validity count and bounding box (REPROJ:207-223) are half-wave shuffle reductions.
// The ordered per-camera person lists (REPROJ:226-233) are built by a flag scan (se3d_k_reproj_scan) and a 428-byte gather
// (se3d_k_reproj_compact).  Everything is fp64 like the node; no FMA contraction (-ffp-contract=off).
#include <cstdlib>

#include "se3d_device.h"

namespace {

constexpr int kCamsPerBlock = 8;  // 8 half-waves of 32 lanes

// Result slot b = blockIdx.z of a batched launch (se3d_reproject_resident_batch: every frame slot of a resident walk in one launch): inputs and
// outputs of slot b are the b-th blocks of the descriptor's arrays.  A single-list call is the batch of one.
__device__ __forceinline__ Se3dReproj se3d_reproj_slot(Se3dReproj R, int b) {
  const size_t nt = (size_t)R.N * R.T;
  R.in += (size_t)b * R.in_stride;
  if (R.n_dev) R.n_dev += b;
  if (R.dense) { R.dense += b * nt; R.valid += b * nt; R.slot += b * nt; }   // (the one-kernel form below has no dense intermediate)
  R.out += b * nt; R.out_src += b * nt;
  R.det_offset += (size_t)b * (R.N + 1);
  return R;
}

// One (person, camera) on a 32-lane half-wave: lane j < 17 projects the 7 sigma points of model joint j (s_X / s_score: drawn once per person),
// weighted mean / 2x2 covariance, in-image gate; count and bounding box by half-wave shuffles.  Returns what lane j publishes for its joint
// and (every lane) the person's count and box.
struct ReprojLane { bool valid; double mx, my, cxx, cxy, cyy; float score; int cnt; double mnx, mny, mxx, mxy; };
__device__ __forceinline__ ReprojLane se3d_reproj_person_cam(const Se3dReproj &R, int cam, int lane, const double (*s_X)[7][3], const float *s_score) {
  const int W = R.wh[2 * cam], Hh = R.wh[2 * cam + 1];
  bool valid = false;
  double mx = 0., my = 0., cxx = 0., cxy = 0., cyy = 0.;
  float score = 0.f;
  if (lane < SE3D_NKP) {
    score = s_score[lane];
    if (score > 0.f) {
      const double w = 1.0 / (2.0 * (3.0 + 0.5));  // weights(0) = 2 kappa / (2 (DIM + kappa)) equals the others for kappa = 0.5
      const double *M = R.cam + 16 * (size_t)cam;
      const double fx = M[12], fy = M[13], pcx = M[14], pcy = M[15];
      double px[7], py[7];
#pragma unroll
      for (int s = 0; s < 7; ++s) {
        const double X0 = s_X[lane][s][0], X1 = s_X[lane][s][1], X2 = s_X[lane][s][2];
        const double xc = (M[0] * X0 + M[1] * X1 + M[2] * X2) + M[3];
        const double yc = (M[4] * X0 + M[5] * X1 + M[6] * X2) + M[7];
        const double zc = (M[8] * X0 + M[9] * X1 + M[10] * X2) + M[11];
        // project3dToPixel (REPROJ:186): (fx X) / Z + cx.  One reciprocal of Z (hardware seed + two Newton steps: <= 1 ulp of fp64) serves both
        // coordinates instead of two IEEE divisions (~45 instructions each, 14 per joint and camera): the pixel differs from the reference's
        // quotient by <= 2 ulp of a double, nine orders of magnitude below the float32 the result is published in
        double iz = __builtin_amdgcn_rcp(zc);
        iz = __builtin_fma(__builtin_fma(-zc, iz, 1.0), iz, iz);
        iz = __builtin_fma(__builtin_fma(-zc, iz, 1.0), iz, iz);
        px[s] = (fx * xc) * iz + pcx;
        py[s] = (fy * yc) * iz + pcy;
      }
#pragma unroll
      for (int s = 0; s < 7; ++s) { mx += px[s] * w; my += py[s] * w; }
#pragma unroll
      for (int s = 0; s < 7; ++s) {
        const double dx = px[s] - mx, dy = py[s] - my;
        cxx += dx * w * dx; cxy += dx * w * dy; cyy += dy * w * dy;
      }
      valid = !(mx < 0. || mx > (double)W || my < 0. || my > (double)Hh) && mx == mx && my == my;  // REPROJ:207 (+ NaN guard)
    }
  }
  // half-wave reductions: count, bbox (REPROJ:153-165, 216-222)
  int cnt = valid ? 1 : 0;
  double mnx = valid ? mx : (double)W, mny = valid ? my : (double)Hh, mxx = valid ? mx : 0., mxy = valid ? my : 0.;
#pragma unroll
  for (int o = 16; o >= 1; o >>= 1) {
    cnt += __shfl_xor(cnt, o, 32);
    mnx = fmin(mnx, __shfl_xor(mnx, o, 32));
    mny = fmin(mny, __shfl_xor(mny, o, 32));
    mxx = fmax(mxx, __shfl_xor(mxx, o, 32));
    mxy = fmax(mxy, __shfl_xor(mxy, o, 32));
  }
  ReprojLane out;
  out.valid = valid; out.mx = mx; out.my = my; out.cxx = cxx; out.cxy = cxy; out.cyy = cyy; out.score = score; out.cnt = cnt;
  out.mnx = mnx; out.mny = mny; out.mxx = mxx; out.mxy = mxy;
  return out;
}

// the 7 sigma points of one joint (REPROJ:62-75: cov.llt(), +- sqrt(3 + kappa) L columns) and its score
__device__ __forceinline__ void se3d_reproj_sigma(const se3d_keypoint3d &kp, double (*X)[3], float &score) {
  score = kp.score;
  if (kp.score > 0.f) {  // REPROJ:167-168
    const double c00 = kp.cov[0], c10 = kp.cov[1], c20 = kp.cov[2], c11 = kp.cov[3], c21 = kp.cov[4], c22 = kp.cov[5];
    const double l00 = sqrt(c00);
    const double l10 = c10 / l00, l20 = c20 / l00;
    const double l11 = sqrt(c11 - l10 * l10);
    const double l21 = (c21 - l20 * l10) / l11;
    const double l22 = sqrt(c22 - l20 * l20 - l21 * l21);
    const double L[3][3] = {{l00, 0., 0.}, {l10, l11, 0.}, {l20, l21, l22}};
    const double b = sqrt(3.0 + 0.5);   // kappa = 0.5
#pragma unroll
    for (int s_ = 0; s_ < 7; ++s_)
#pragma unroll
      for (int a = 0; a < 3; ++a)
        X[s_][a] = s_ == 0 ? kp.joint[a] : (s_ < 4 ? L[a][s_ - 1] * (-b) + kp.joint[a] : L[a][s_ - 4] * b + kp.joint[a]);
  }
}
__device__ __forceinline__ void se3d_reproj_store(se3d_person2d &o, const ReprojLane &r, int lane) {
  if (lane < SE3D_NKP) {
    se3d_keypoint2d k;
    if (r.valid) {
      k.x = (float)r.mx; k.y = (float)r.my; k.score = r.score;
      k.cov[0] = (float)r.cxx; k.cov[1] = (float)r.cxy; k.cov[2] = (float)r.cyy;
    } else {
      k.x = k.y = k.score = 0.f; k.cov[0] = k.cov[1] = k.cov[2] = 0.f;
    }
    o.keypoints[lane] = k;
  } else if (lane == SE3D_NKP) {
    o.score = 1.0f;  // REPROJ:154
    o.bbox[0] = (float)r.mnx; o.bbox[1] = (float)r.mny; o.bbox[2] = (float)r.mxx; o.bbox[3] = (float)r.mxy;
  }
}

__global__ __launch_bounds__(256) void se3d_k_reproject(Se3dReproj Rb) {
  const Se3dReproj R = se3d_reproj_slot(Rb, blockIdx.z);
  const int lane = threadIdx.x & 31;
  const int cam = blockIdx.x * kCamsPerBlock + (threadIdx.x >> 5);
  const int person = blockIdx.y;
  const int n = R.n_dev ? *R.n_dev : R.n;
  if (person >= n) return;   // block-uniform
  // The sigma points of a joint depend on the person only: the first half-wave draws them once per block into LDS and the eight (person,
  // camera) half-waves of the block project them — round 1 redid the 3x3 Cholesky (3 square roots, 3 divisions in fp64) for every camera.
  __shared__ double s_X[SE3D_NKP][7][3];
  __shared__ float s_score[SE3D_NKP];
  if (threadIdx.x < SE3D_NKP) se3d_reproj_sigma(R.in[person].keypoints[c_kp2fusion[R.pose_method][threadIdx.x]], s_X[threadIdx.x], s_score[threadIdx.x]);
  __syncthreads();
  if (cam >= R.N) return;
  const ReprojLane r = se3d_reproj_person_cam(R, cam, lane, s_X, s_score);
  se3d_reproj_store(R.dense[(size_t)cam * R.T + person], r, lane);
  if (lane == SE3D_NKP) R.valid[(size_t)person * R.N + cam] = r.cnt > 0 ? 1 : 0;
}

// Small rigs (round 3): ONE workgroup per result slot does everything — no dense [N][T] intermediate in HBM (707 MB written and 242 MB read
// back per 16 384-slot launch at 16 cameras), no scan and compaction launches.  1024 threads = 32 half-waves = 32 / N persons x N cameras per
// trip; the projected persons of the slot wait in LDS ([n][N] x 428 B) until all are done, then the per-camera counts are scanned by one
// wavefront and a wavefront per (person, camera) pair copies the valid ones to their list slots (camera-major, person order: REPROJ:226-233).
// Same arithmetic as se3d_k_reproject (shared device functions) -> identical lists.
__global__ __launch_bounds__(1024) void se3d_k_reproject_lists(Se3dReproj Rb) {
  const Se3dReproj R = se3d_reproj_slot(Rb, blockIdx.z);
  extern __shared__ __align__(16) unsigned char rl_smem[];
  const int N = R.N, T = R.T;
  se3d_person2d *res = reinterpret_cast<se3d_person2d *>(rl_smem);                    // [T][N]
  double (*s_X)[SE3D_NKP][7][3] = reinterpret_cast<double (*)[SE3D_NKP][7][3]>(rl_smem + ((sizeof(se3d_person2d) * (size_t)T * N + 15) & ~(size_t)15));   // [ppr][17][7][3]
  const int ppr = 32 / N;                                                             // persons per trip
  float (*s_score)[SE3D_NKP] = reinterpret_cast<float (*)[SE3D_NKP]>(reinterpret_cast<unsigned char *>(s_X) + sizeof(double) * (size_t)ppr * SE3D_NKP * 21);
  unsigned char *s_valid = reinterpret_cast<unsigned char *>(s_score) + sizeof(float) * (size_t)ppr * SE3D_NKP;   // [T][N]
  __shared__ int s_off[64];
  const int tid = threadIdx.x, lane = tid & 31, hw = tid >> 5;
  const int n = min(R.n_dev ? *R.n_dev : R.n, T);
  const int pi = hw / N, cam = hw - pi * N;
  for (int p0 = 0; p0 < n; p0 += ppr) {
    if (tid < ppr * SE3D_NKP) {
      const int pp = tid / SE3D_NKP, j = tid - pp * SE3D_NKP;
      if (p0 + pp < n) se3d_reproj_sigma(R.in[p0 + pp].keypoints[c_kp2fusion[R.pose_method][j]], s_X[pp][j], s_score[pp][j]);
    }
    __syncthreads();
    const int person = p0 + pi;
    if (pi < ppr && person < n) {
      const ReprojLane r = se3d_reproj_person_cam(R, cam, lane, s_X[pi], s_score[pi]);
      se3d_reproj_store(res[(size_t)person * N + cam], r, lane);
      if (lane == SE3D_NKP) s_valid[person * N + cam] = r.cnt > 0 ? 1 : 0;
    }
    __syncthreads();
  }
  if (tid < 64) {   // per-camera counts -> exclusive offsets (N <= 32)
    int cnt = 0;
    if (tid < N)
      for (int p = 0; p < n; ++p) cnt += s_valid[p * N + tid];
    int inc = cnt;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (tid >= o) inc += t; }
    s_off[tid] = inc - cnt;
    if (tid < N) R.det_offset[tid + 1] = inc;
    if (tid == 0) R.det_offset[0] = 0;
  }
  __syncthreads();
  constexpr int kWords = sizeof(se3d_person2d) / 4;
  const int l64 = tid & 63, wave = tid >> 6;
  for (int pair = wave; pair < n * N; pair += 16) {
    const int person = pair / N, c = pair - person * N;
    if (!s_valid[pair]) continue;
    int rank = 0;
    for (int p = 0; p < person; ++p) rank += s_valid[p * N + c];
    const int slot = s_off[c] + rank;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(&res[pair]);
    uint32_t *dst = reinterpret_cast<uint32_t *>(&R.out[slot]);
    dst[l64] = src[l64];
    if (l64 + 64 < kWords) dst[l64 + 64] = src[l64 + 64];
    if (l64 == 0) R.out_src[slot] = person;
  }
}

// one block, thread c = camera c: per-camera counts -> exclusive offsets -> output slot of every (person, camera)
__global__ __launch_bounds__(256) void se3d_k_reproj_scan(Se3dReproj Rb) {
  const Se3dReproj R = se3d_reproj_slot(Rb, blockIdx.z);
  __shared__ int s_cnt[256];
  const int c = threadIdx.x;
  const int n = R.n_dev ? *R.n_dev : R.n;
  int cnt = 0;
  if (c < R.N)
    for (int p = 0; p < n; ++p) cnt += R.valid[(size_t)p * R.N + c];
  s_cnt[c] = cnt;
  __syncthreads();
  for (int o = 1; o < 256; o <<= 1) {  // inclusive Hillis-Steele scan
    const int v = (c >= o) ? s_cnt[c - o] : 0;
    __syncthreads();
    s_cnt[c] += v;
    __syncthreads();
  }
  if (c < R.N) {
    int off = s_cnt[c] - cnt;
    if (c == 0) R.det_offset[0] = 0;
    R.det_offset[c + 1] = s_cnt[c];
    for (int p = 0; p < n; ++p) {
      const bool v = R.valid[(size_t)p * R.N + c] != 0;
      R.slot[(size_t)p * R.N + c] = v ? off : -1;
      off += v ? 1 : 0;
    }
  }
}

// one workgroup per result slot: a wavefront per (person, camera) pair in turn copies the 107 dwords of a valid projected person to its list
// slot.  (Round 1 launched a block per two pairs: at 16 384 slots per launch that was a million workgroups, most of them exiting at once,
// 0.80 ms of workgroup dispatch for 0.45 GB of copies.)
__global__ __launch_bounds__(256) void se3d_k_reproj_compact(Se3dReproj Rb) {
  const Se3dReproj R = se3d_reproj_slot(Rb, blockIdx.z);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = R.n_dev ? *R.n_dev : R.n;
  constexpr int kWords = sizeof(se3d_person2d) / 4;
  for (int pair = blockIdx.x * 4 + wave; pair < n * R.N; pair += 4 * gridDim.x) {
    const int person = pair / R.N, cam = pair - person * R.N;
    const int slot = R.slot[(size_t)person * R.N + cam];
    if (slot < 0) continue;
    const uint32_t *src = reinterpret_cast<const uint32_t *>(&R.dense[(size_t)cam * R.T + person]);
    uint32_t *dst = reinterpret_cast<uint32_t *>(&R.out[slot]);
    const uint32_t a = src[lane], b = lane + 64 < kWords ? src[lane + 64] : 0u;
    dst[lane] = a;
    if (lane + 64 < kWords) dst[lane + 64] = b;
    if (lane == 0) R.out_src[slot] = person;
  }
}

}  // namespace

// dynamic LDS of the one-kernel form for a rig of N cameras and T person slots, 0 when the three-kernel form is used
size_t se3d_reproj_lists_lds(int N, int T, int dense_form) {
  if (N > 32 || dense_form) return 0;
  const size_t b = ((sizeof(se3d_person2d) * (size_t)T * N + 15) & ~(size_t)15) + (size_t)(32 / N) * SE3D_NKP * (21 * sizeof(double) + sizeof(float)) + (size_t)T * N + 16;
  return b <= 96 * 1024 ? b : 0;
}

// Kernel attributes are per DEVICE in HIP: called by se3d_create for the context's device (after hipSetDevice), like se3d_stage1_init_device.
extern "C" hipError_t se3d_reproj_init_device(void) {
  return hipFuncSetAttribute(reinterpret_cast<const void *>(se3d_k_reproject_lists), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
}

hipError_t se3d_launch_reproject(const Se3dReproj &R, int n_grid, hipStream_t stream, int batch) {
  if (n_grid <= 0 || batch <= 0) {
    return hipMemsetAsync(R.det_offset, 0, sizeof(int32_t) * (R.N + 1) * (size_t)(batch > 0 ? batch : 1), stream);
  }
  // small rigs: one workgroup per result slot, lists written directly (options.reproj_form = 1: the three-kernel form)
  const size_t lds_lists = se3d_reproj_lists_lds(R.N, R.T, R.dense_form);
  if (lds_lists > 0) {
    // (the kernel's dynamic-LDS limit is raised per device by se3d_reproj_init_device, called from se3d_create)
    hipLaunchKernelGGL(se3d_k_reproject_lists, dim3(1, 1, batch), dim3(1024), lds_lists, stream, R);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(se3d_k_reproject, dim3((R.N + kCamsPerBlock - 1) / kCamsPerBlock, n_grid, batch), dim3(256), 0, stream, R);
  hipLaunchKernelGGL(se3d_k_reproj_scan, dim3(1, 1, batch), dim3(256), 0, stream, R);
  // few slots: spread a slot's pairs over several workgroups; many slots: one workgroup each
  const int cgrid = batch >= 1024 ? 1 : (n_grid * R.N + 3) / 4 < 256 ? (n_grid * R.N + 3) / 4 : 256;
  hipLaunchKernelGGL(se3d_k_reproj_compact, dim3(cgrid, 1, batch), dim3(256), 0, stream, R);
  return hipGetLastError();
}
