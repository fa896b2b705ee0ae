// se3d_oracle.cpp — CPU ORACLE (TEST INFRASTRUCTURE ONLY; see se3d_oracle.h for the pinning status).
//
// Restates, in dependency-free C++17, the algorithm of the reference hot path.  Citations:
//   S3D = skeleton_3d/src/skeleton_3d_triang_mult_node.cpp   PP = pose_prior/src/pose_prior_mult_node.cpp
//   HUN = skeleton_3d/src/Hungarian.cpp                      FBP = skeleton_3d/include/skeleton_3d/fusion_body_parts.h
// Float/double types follow the reference exactly; float expressions are written with the association order the
// reference's Eigen 3.3 expressions evaluate to (fixed-size redux of 3: a0+(a1+a2); of 4: (a0+a1)+(a2+a3)).
// Build with -ffp-contract=off (the catkin Release build of the reference sets no -march, hence no FMA).
#include "se3d_oracle.h"

#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include <atomic>

namespace {

// Branch counters (se3d_oracle_stats_*): every data-dependent branch of the reference that a test must PROVE it exercised bumps one.
std::atomic<long long> g_stat[SE3D_ORACLE_STAT_COUNT];
inline void stat(int i, long long n = 1) { g_stat[i].fetch_add(n, std::memory_order_relaxed); }

// ------------------------------------------------------------------------------------------------------------
// constants (row K)
// ------------------------------------------------------------------------------------------------------------
constexpr int NKP = SE3D_NUM_KEYPOINTS;
constexpr int NFU = SE3D_NUM_FUSION_KEYPOINTS;
constexpr double MAX_COSTS = 1e6;                          // S3D:43
constexpr int g_min_num_valid_keypoints = 9;               // S3D:57
constexpr float g_triangulation_threshold = 0.30f;         // S3D:58,1099
constexpr double g_reproj_error_max_acceptable = 0.050;    // S3D:59
constexpr double g_max_joint_dist_to_root = 2.0;           // S3D:61
constexpr double g_merge_dist_thresh = 0.20;               // S3D:62, PP:64
constexpr double g_limbLCovOffsetSigma = 0.075;            // S3D:149
// pose_prior node constants (PP:46-66)
constexpr float g_min_score = 0.10f;                       // PP:50
constexpr double g_avg_delay = 0.10, g_root_sigma_factor = 100.0, g_predNoiseSigma = 0.12, g_defaultResSigma = 0.10;   // PP:51, 52, 47, 48
constexpr int g_n_mov_avg = 3, g_min_num_obs_track = 10;  // PP:53, 66
constexpr double g_t_max_unobserved = 1.0, g_dist_threshold = 5.0;   // PP:62, 63

// FUSION_BODY_PARTS (FBP:3-24)
enum Fusion {
  F_Nose = 0, F_Neck = 1, F_RShoulder = 2, F_RElbow = 3, F_RWrist = 4, F_LShoulder = 5, F_LElbow = 6, F_LWrist = 7,
  F_MidHip = 8, F_RHip = 9, F_RKnee = 10, F_RAnkle = 11, F_LHip = 12, F_LKnee = 13, F_LAnkle = 14, F_REye = 15,
  F_LEye = 16, F_REar = 17, F_LEar = 18, F_Head = 19, F_Belly = 20
};
constexpr double vel_sigmas[NFU] = {2., 1., 1., 2., 3., 1., 2., 3., 1., 1., 2., 3., 1., 2., 3., 2., 2., 2., 2., 2., 1.};  // FBP:33

// EdgeTPU_BodyParts_Simple (S3D:81-104)
constexpr int simple_kpParent[NKP] = {-1, 0, 0, 1, 2, 0, 0, 5, 6, 7, 8, 5, 6, 11, 12, 13, 14};
constexpr double simple_limbLength[NKP] = {-1, 0.05, 0.05, 0.10, 0.10, -1, -1, 0.28, 0.28, 0.25, 0.25, 0.50, 0.50, 0.45, 0.45, 0.446, 0.446};
constexpr double simple_limbLSigma[NKP] = {-1, 0.05, 0.05, 0.05, 0.05, -1, -1, 0.10, 0.10, 0.10, 0.10, 0.15, 0.15, 0.10, 0.10, 0.10, 0.10};
constexpr double simple_shoulderDist = 0.35, simple_shoulderSigma = 0.15;
constexpr int SIMPLE_RShoulder = 6, SIMPLE_LShoulder = 5;
// EdgeTPU_BodyParts_H36M (S3D:111-133)
constexpr int h36m_kpParent[NKP] = {-1, 0, 0, 2, 3, 2, 2, 5, 6, 7, 8, 4, 4, 11, 12, 13, 14};
constexpr double h36m_limbLength[NKP] = {-1, 0.115, 0.116, 0.255, 0.238, 0.149, 0.149, 0.28, 0.28, 0.25, 0.25, 0.134, 0.134, 0.449, 0.449, 0.446, 0.446};
constexpr double h36m_limbLSigma[NKP] = {-1, 0.07, 0.07, 0.15, 0.15, 0.10, 0.10, 0.15, 0.15, 0.15, 0.15, 0.10, 0.10, 0.20, 0.20, 0.20, 0.20};
// g_kp2kpFusion_idx_* (S3D:139-145)
constexpr int kp2fusion_simple[NKP] = {F_Nose, F_LEye, F_REye, F_LEar, F_REar, F_LShoulder, F_RShoulder, F_LElbow, F_RElbow,
                                       F_LWrist, F_RWrist, F_LHip, F_RHip, F_LKnee, F_RKnee, F_LAnkle, F_RAnkle};
constexpr int kp2fusion_h36m[NKP] = {F_Nose, F_Head, F_Neck, F_Belly, F_MidHip, F_LShoulder, F_RShoulder, F_LElbow, F_RElbow,
                                     F_LWrist, F_RWrist, F_LHip, F_RHip, F_LKnee, F_RKnee, F_LAnkle, F_RAnkle};

struct ModelTables {
  const int *kpParent;
  const double *limbLength;
  const double *limbLSigma;
  const int *kp2fusion;
  bool simple;
};
ModelTables tables_for(int pose_method) {  // S3D:1101-1112
  if (pose_method == SE3D_POSE_H36M) return {h36m_kpParent, h36m_limbLength, h36m_limbLSigma, kp2fusion_h36m, false};
  return {simple_kpParent, simple_limbLength, simple_limbLSigma, kp2fusion_simple, true};
}

struct V3f { float x, y, z; };
struct M2f { float a00, a01, a10, a11; };

// ------------------------------------------------------------------------------------------------------------
// small dense linear algebra (restating the Eigen routines the reference calls)
// ------------------------------------------------------------------------------------------------------------

// One-sided (Hestenes) Jacobi SVD in double of an m x n matrix (m >= n), column-major a[i + m*j].
// Used only for the 3x4 pseudo-inverse of the rig set-up (S3D:236-240), where the reference uses JacobiSVD<double>.
void hestenes_svd(int m, int n, std::vector<double> &a, std::vector<double> &v, std::vector<double> &s) {
  v.assign(n * n, 0.0);
  for (int i = 0; i < n; ++i) v[i + n * i] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int i = 0; i < m; ++i) {
          alpha += a[i + m * p] * a[i + m * p];
          beta += a[i + m * q] * a[i + m * q];
          gamma += a[i + m * p] * a[i + m * q];
        }
        if (gamma == 0.0) continue;
        off = std::max(off, std::fabs(gamma) / std::sqrt(alpha * beta));
        double zeta = (beta - alpha) / (2.0 * gamma);
        double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
        double c = 1.0 / std::sqrt(1.0 + t * t), sn = c * t;
        for (int i = 0; i < m; ++i) {
          double x = a[i + m * p], y = a[i + m * q];
          a[i + m * p] = c * x - sn * y;
          a[i + m * q] = sn * x + c * y;
        }
        for (int i = 0; i < n; ++i) {
          double x = v[i + n * p], y = v[i + n * q];
          v[i + n * p] = c * x - sn * y;
          v[i + n * q] = sn * x + c * y;
        }
      }
    if (off < 1e-16) break;
  }
  s.assign(n, 0.0);
  for (int j = 0; j < n; ++j) {
    double nn = 0;
    for (int i = 0; i < m; ++i) nn += a[i + m * j] * a[i + m * j];
    s[j] = std::sqrt(nn);
    if (s[j] > 0)
      for (int i = 0; i < m; ++i) a[i + m * j] /= s[j];  // a now holds U
  }
}

// pseudo_inv34d (S3D:236-240): P (3x4 row-major) -> Pinv (4x3 row-major)
void pseudo_inv34d(const double P[12], double Pinv[12]) {
  // SVD of P^T (4x3): P^T = U S V^T  ->  P = V S U^T, pinv(P) = U S^+ V^T
  std::vector<double> a(12), v, s;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 3; ++j) a[i + 4 * j] = P[j * 4 + i];
  hestenes_svd(4, 3, a, v, s);
  double smax = std::max(s[0], std::max(s[1], s[2]));
  double tol = std::numeric_limits<double>::epsilon() * 4.0 * smax;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 3; ++j) {
      double acc = 0;
      for (int k = 0; k < 3; ++k) {
        double sinv = (std::fabs(s[k]) > tol) ? 1.0 / s[k] : 0.0;
        acc += a[i + 4 * k] * sinv * v[j + 3 * k];
      }
      Pinv[i * 3 + j] = acc;
    }
}

// Null vector of a (rows x 4) float matrix: restatement of Eigen 3.3 JacobiSVD<Matrix<float,-1,4>>(ComputeThinV)
// .matrixV().col(3) as called at S3D:456 — for rows != 4 the ColPivHouseholderQR preconditioner to a 4x4 upper-triangular R
// (JacobiSVD::compute step 1, `if (m_rows != m_cols)`); for rows == 4 (a joint seen by exactly TWO views) Eigen skips the
// preconditioner: the work matrix is the scaled input itself and V starts as the identity; then two-sided
// Jacobi sweeps on the work matrix (real_2x2_jacobi_svd), descending sort of the singular values, last column of V.
// All arithmetic in float like the reference.  Inner products are plain sequential sums (Eigen's vectorised
// summation order is not reproduced: "parity unpinned", see header).
struct JacobiRot { float c, s; };

inline bool make_jacobi(float x, float y, float z, JacobiRot &r) {  // JacobiRotation::makeJacobi
  float deno = 2.0f * std::fabs(y);
  if (deno < std::numeric_limits<float>::min()) { r.c = 1.f; r.s = 0.f; return false; }
  float tau = (x - z) / deno;
  float w = std::sqrt(tau * tau + 1.0f);
  float t = (tau > 0.f) ? 1.0f / (tau + w) : 1.0f / (tau - w);
  float sign_t = t > 0.f ? 1.f : -1.f;
  float n = 1.0f / std::sqrt(t * t + 1.0f);
  r.s = -sign_t * (y / std::fabs(y)) * std::fabs(t) * n;
  r.c = n;
  return true;
}

// JacobiSVD::compute steps 2-4 on the 4x4 work matrix W with the accumulated right factor V; v_out = V.col(3) after the descending sort.
void jacobi_null_vector_4x4(float W[4][4], float V[4][4], float v_out[4]) {
  // --- two-sided Jacobi on W (JacobiSVD::compute step 2) ---
  const float precision = 2.0f * std::numeric_limits<float>::epsilon();
  const float considerAsZero = std::numeric_limits<float>::min();
  float maxDiag = 0.f;
  for (int i = 0; i < 4; ++i) maxDiag = std::max(maxDiag, std::fabs(W[i][i]));
  bool finished = false;
  int guard = 0;
  while (!finished && guard++ < 200) {
    finished = true;
    for (int p = 1; p < 4; ++p)
      for (int q = 0; q < p; ++q) {
        float threshold = std::max(considerAsZero, precision * maxDiag);
        if (std::fabs(W[p][q]) > threshold || std::fabs(W[q][p]) > threshold) {
          finished = false;
          // real_2x2_jacobi_svd
          float m00 = W[p][p], m01 = W[p][q], m10 = W[q][p], m11 = W[q][q];
          JacobiRot rot1;
          float t = m00 + m11, d = m10 - m01;
          if (std::fabs(d) < std::numeric_limits<float>::min()) {
            rot1.s = 0.f; rot1.c = 1.f;
          } else {
            float u = t / d;
            float tm = std::sqrt(1.0f + u * u);
            rot1.s = 1.0f / tm;
            rot1.c = u / tm;
          }
          // m.applyOnTheLeft(0,1,rot1): x' = c x + s y ; y' = -s x + c y  (rows)
          float n00 = rot1.c * m00 + rot1.s * m10, n01 = rot1.c * m01 + rot1.s * m11;
          float n10 = -rot1.s * m00 + rot1.c * m10, n11 = -rot1.s * m01 + rot1.c * m11;
          (void)n10;
          JacobiRot jr;
          make_jacobi(n00, n01, n11, jr);
          // j_left = rot1 * j_right.transpose();  transpose = (c, -s);  product (c1*c2 - s1*s2, c1*s2 + s1*c2)
          JacobiRot jrt{jr.c, -jr.s};
          JacobiRot jl{rot1.c * jrt.c - rot1.s * jrt.s, rot1.c * jrt.s + rot1.s * jrt.c};
          // W.applyOnTheLeft(p,q,jl)
          for (int j = 0; j < 4; ++j) {
            float x = W[p][j], y = W[q][j];
            W[p][j] = jl.c * x + jl.s * y;
            W[q][j] = -jl.s * x + jl.c * y;
          }
          // W.applyOnTheRight(p,q,jr): x' = c x - s y ; y' = s x + c y (columns)
          for (int i = 0; i < 4; ++i) {
            float x = W[i][p], y = W[i][q];
            W[i][p] = jr.c * x - jr.s * y;
            W[i][q] = jr.s * x + jr.c * y;
          }
          for (int i = 0; i < 4; ++i) {
            float x = V[i][p], y = V[i][q];
            V[i][p] = jr.c * x - jr.s * y;
            V[i][q] = jr.s * x + jr.c * y;
          }
          maxDiag = std::max(maxDiag, std::max(std::fabs(W[p][p]), std::fabs(W[q][q])));
        }
      }
  }
  // singular values, descending selection sort with column swaps (step 4) — only the smallest matters
  float sv[4];
  int idx[4] = {0, 1, 2, 3};
  for (int i = 0; i < 4; ++i) sv[i] = std::fabs(W[i][i]);
  for (int i = 0; i < 4; ++i) {
    int pos = i;
    for (int j = i + 1; j < 4; ++j)
      if (sv[j] > sv[pos]) pos = j;
    if (sv[pos] == 0.f) break;
    if (pos != i) { std::swap(sv[i], sv[pos]); std::swap(idx[i], idx[pos]); }
  }
  for (int i = 0; i < 4; ++i) v_out[i] = V[i][idx[3]];
}

void null_vector_f32(int rows, std::vector<float> &A /*col-major rows x 4*/, float v_out[4]) {
  const int cols = 4;
  // scale (JacobiSVD::compute)
  float scale = 0.f;
  for (float x : A) scale = std::max(scale, std::fabs(x));
  if (scale == 0.f) scale = 1.f;
  for (float &x : A) x = x / scale;
  float W[4][4], V[4][4];
  if (rows == cols) {
    // square input: no QR step (Eigen 3.3 JacobiSVD::compute: m_workMatrix = matrix / scale, m_matrixV.setIdentity())
    for (int i = 0; i < 4; ++i)
      for (int j = 0; j < 4; ++j) {
        W[i][j] = A[i + rows * j];
        V[i][j] = (i == j) ? 1.f : 0.f;
      }
    jacobi_null_vector_4x4(W, V, v_out);
    return;
  }
  // --- ColPivHouseholderQR ---
  int perm[4] = {0, 1, 2, 3};
  float normsUpdated[4], normsDirect[4];
  for (int k = 0; k < cols; ++k) {
    float ss = 0.f;
    for (int i = 0; i < rows; ++i) ss += A[i + rows * k] * A[i + rows * k];
    normsUpdated[k] = normsDirect[k] = std::sqrt(ss);
  }
  const float norm_downdate_threshold = std::sqrt(std::numeric_limits<float>::epsilon());
  int size = std::min(rows, cols);
  std::vector<float> tmp(cols);
  for (int k = 0; k < size; ++k) {
    int big = k;
    for (int j = k + 1; j < cols; ++j)
      if (normsUpdated[j] > normsUpdated[big]) big = j;
    if (big != k) {
      for (int i = 0; i < rows; ++i) std::swap(A[i + rows * k], A[i + rows * big]);
      std::swap(normsUpdated[k], normsUpdated[big]);
      std::swap(normsDirect[k], normsDirect[big]);
      std::swap(perm[k], perm[big]);
    }
    // makeHouseholderInPlace on A.col(k).tail(rows-k)
    float tailSq = 0.f;
    for (int i = k + 1; i < rows; ++i) tailSq += A[i + rows * k] * A[i + rows * k];
    float c0 = A[k + rows * k], tau, beta;
    if (tailSq <= std::numeric_limits<float>::min()) {
      tau = 0.f;
      beta = c0;
      for (int i = k + 1; i < rows; ++i) A[i + rows * k] = 0.f;
    } else {
      beta = std::sqrt(c0 * c0 + tailSq);
      if (c0 >= 0.f) beta = -beta;
      for (int i = k + 1; i < rows; ++i) A[i + rows * k] = A[i + rows * k] / (c0 - beta);
      tau = (beta - c0) / beta;
    }
    A[k + rows * k] = beta;
    // applyHouseholderOnTheLeft to bottomRightCorner(rows-k, cols-k-1)
    if (tau != 0.f && k + 1 < cols) {
      if (rows - k == 1) {
        for (int j = k + 1; j < cols; ++j) A[k + rows * j] *= (1.0f - tau);
      } else {
        for (int j = k + 1; j < cols; ++j) {
          float t = 0.f;
          for (int i = k + 1; i < rows; ++i) t += A[i + rows * k] * A[i + rows * j];
          t += A[k + rows * j];
          tmp[j] = t;
        }
        for (int j = k + 1; j < cols; ++j) {
          A[k + rows * j] -= tau * tmp[j];
          for (int i = k + 1; i < rows; ++i) A[i + rows * j] -= tau * A[i + rows * k] * tmp[j];
        }
      }
    }
    // LAPACK-style column-norm downdate (lawn176)
    for (int j = k + 1; j < cols; ++j) {
      if (normsUpdated[j] != 0.f) {
        float temp = std::fabs(A[k + rows * j]) / normsUpdated[j];
        temp = (1.0f + temp) * (1.0f - temp);
        temp = temp < 0.f ? 0.f : temp;
        float r = normsUpdated[j] / normsDirect[j];
        float temp2 = temp * (r * r);
        if (temp2 <= norm_downdate_threshold) {
          float ss = 0.f;
          for (int i = k + 1; i < rows; ++i) ss += A[i + rows * j] * A[i + rows * j];
          normsDirect[j] = std::sqrt(ss);
          normsUpdated[j] = normsDirect[j];
        } else {
          normsUpdated[j] *= std::sqrt(temp);
        }
      }
    }
  }
  // work matrix W = R (4x4 upper triangular), V = column permutation
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      W[i][j] = (j >= i && i < rows) ? A[i + rows * j] : 0.f;
      V[i][j] = 0.f;
    }
  for (int j = 0; j < 4; ++j) V[perm[j]][j] = 1.f;
  jacobi_null_vector_4x4(W, V, v_out);
}

// ------------------------------------------------------------------------------------------------------------
// row M — Munkres / Hungarian (HUN:60-397), restated with index arrays instead of boolean matrices.
// Each row/column holds at most one star and each row at most one prime between clears, so starCol[row],
// starRow[col], primeCol[row] carry the same information as starMatrix/primeMatrix; every scan keeps the
// reference's order (HUN:122-130, 157-166, 278-304) so ties resolve identically.
// ------------------------------------------------------------------------------------------------------------
void hungarian(int *assignment, double *cost, const double *distIn, int nR, int nC) {
  *cost = 0;
  for (int r = 0; r < nR; ++r) assignment[r] = -1;
  if (nR <= 0 || nC <= 0) return;
  std::vector<double> d(distIn, distIn + (size_t)nR * nC);
  std::vector<char> covC(nC, 0), covR(nR, 0);
  std::vector<int> starCol(nR, -1), starRow(nC, -1), primeCol(nR, -1);
  int minDim;
  auto D = [&](int r, int c) -> double & { return d[r + (size_t)nR * c]; };
  auto isZero = [&](int r, int c) { return std::fabs(D(r, c)) < DBL_EPSILON; };
  if (nR <= nC) {  // HUN:94-131
    minDim = nR;
    for (int r = 0; r < nR; ++r) {
      double mn = D(r, 0);
      for (int c = 1; c < nC; ++c) mn = (D(r, c) < mn) ? D(r, c) : mn;
      for (int c = 0; c < nC; ++c) D(r, c) -= mn;
    }
    for (int r = 0; r < nR; ++r)
      for (int c = 0; c < nC; ++c)
        if (isZero(r, c) && !covC[c]) {
          starCol[r] = c; starRow[c] = r; covC[c] = 1;
          break;
        }
  } else {  // HUN:132-170
    minDim = nC;
    for (int c = 0; c < nC; ++c) {
      double mn = D(0, c);
      for (int r = 1; r < nR; ++r) mn = (D(r, c) < mn) ? D(r, c) : mn;
      for (int r = 0; r < nR; ++r) D(r, c) -= mn;
    }
    for (int c = 0; c < nC; ++c)
      for (int r = 0; r < nR; ++r)
        if (isZero(r, c) && !covR[r]) {
          starCol[r] = c; starRow[c] = r; covC[c] = 1; covR[r] = 1;
          break;
        }
    std::fill(covR.begin(), covR.end(), 0);
  }
  // iterative form of the mutually tail-recursive steps 2b -> 3 -> (4 -> 2a -> 2b | 5 -> 3)
  enum { S2B, S3, S5, DONE } state = S2B;
  while (state != DONE) {
    if (state == S2B) {  // HUN:245-266
      int n = 0;
      for (int c = 0; c < nC; ++c) n += covC[c] ? 1 : 0;
      state = (n == minDim) ? DONE : S3;
    } else if (state == S3) {  // HUN:269-309
      bool zerosFound = true, toStep4 = false;
      int r4 = -1, c4 = -1;
      while (zerosFound && !toStep4) {
        zerosFound = false;
        for (int c = 0; c < nC && !toStep4; ++c) {
          if (covC[c]) continue;
          for (int r = 0; r < nR; ++r) {
            if (!covR[r] && isZero(r, c)) {
              primeCol[r] = c;
              int sc = starCol[r];
              if (sc < 0) { toStep4 = true; r4 = r; c4 = c; }
              else { covR[r] = 1; covC[sc] = 0; zerosFound = true; }
              break;
            }
          }
        }
      }
      if (toStep4) {  // step 4 (HUN:312-363): augment along the alternating path, then 2a
        int row = r4, col = c4;
        for (;;) {
          int sr = starRow[col];      // starred zero in current column (old stars)
          starCol[row] = col;         // star the primed zero
          starRow[col] = row;
          if (sr < 0) break;
          int pc = primeCol[sr];      // primed zero in the row of the old star
          row = sr;
          col = pc;                   // old star (sr, oldcol) is un-starred by being overwritten below
        }
        // rebuild starRow consistently (a row's old star column may have been released)
        std::fill(starRow.begin(), starRow.end(), -1);
        for (int r = 0; r < nR; ++r)
          if (starCol[r] >= 0) starRow[starCol[r]] = r;
        std::fill(primeCol.begin(), primeCol.end(), -1);
        std::fill(covR.begin(), covR.end(), 0);
        for (int c = 0; c < nC; ++c)  // step 2a (HUN:221-242): cover every column containing a star
          if (starRow[c] >= 0) covC[c] = 1;
        state = S2B;
      } else {
        state = S5;
      }
    } else {  // S5 (HUN:366-397)
      double h = DBL_MAX;
      for (int r = 0; r < nR; ++r)
        if (!covR[r])
          for (int c = 0; c < nC; ++c)
            if (!covC[c]) { double v = D(r, c); if (v < h) h = v; }
      for (int r = 0; r < nR; ++r)
        if (covR[r])
          for (int c = 0; c < nC; ++c) D(r, c) += h;
      for (int c = 0; c < nC; ++c)
        if (!covC[c])
          for (int r = 0; r < nR; ++r) D(r, c) -= h;
      state = S3;
    }
  }
  for (int r = 0; r < nR; ++r) assignment[r] = starCol[r];  // buildassignmentvector (HUN:190-205)
  for (int r = 0; r < nR; ++r)                               // computeassignmentcost (HUN:208-218)
    if (assignment[r] >= 0) *cost += distIn[r + (size_t)nR * assignment[r]];
}

// ------------------------------------------------------------------------------------------------------------
// stage 1 (skeleton_3d)
// ------------------------------------------------------------------------------------------------------------
int get_fundamental_idx(int i, int j, int NUM_CAMERAS) {  // S3D:242-253
  if (i >= j) return -1;
  if (i > NUM_CAMERAS - 2 || j > NUM_CAMERAS - 1) return -1;
  int start_idx = 0;
  for (int ii = 0; ii < i; ++ii) start_idx += NUM_CAMERAS - ii - 1;
  return start_idx + j - i - 1;
}

// normalize_keypoints (S3D:312-333)
int normalize_keypoints(std::array<V3f, NKP> &kps, std::array<M2f, NKP> &covs, const float K[4], const se3d_person2d &human) {
  int num_valid = 0;
  const float fx = K[0], fy = K[1], cx = K[2], cy = K[3];
  for (int k = 0; k < NKP; ++k) {
    const se3d_keypoint2d &kp = human.keypoints[k];
    if (kp.score >= g_triangulation_threshold) {
      kps[k] = V3f{(kp.x - cx) / fx, (kp.y - cy) / fy, kp.score};
      covs[k] = M2f{kp.cov[0] / (fx * fx), kp.cov[1] / (fx * fy), kp.cov[1] / (fx * fy), kp.cov[2] / (fy * fy)};
      ++num_valid;
    }
  }
  return num_valid;
}

// inner joint loop of calcCost for one view pair (S3D:353-365): F is row-major, index (row*3+col)
inline double pair_cost(const float *F, const V3f *hyp_kps, const V3f *det_kps, int &n_joints_used) {
  double cost = 0.;
  n_joints_used = 0;
  for (int k = 0; k < NKP; ++k) {
    if (hyp_kps[k].z > g_triangulation_threshold && det_kps[k].z > g_triangulation_threshold) {
      const float p1x = hyp_kps[k].x, p1y = hyp_kps[k].y;  // p1 = (x, y, 1)
      const float p2x = det_kps[k].x, p2y = det_kps[k].y;  // p2 = (x, y, 1)
      // epipolar_line_1 = F * p1 ; coefficient i = F(i,0)*p1x + (F(i,1)*p1y + F(i,2)*1)
      const float l1x = F[0] * p1x + (F[1] * p1y + F[2]);
      const float l1y = F[3] * p1x + (F[4] * p1y + F[5]);
      const float l1z = F[6] * p1x + (F[7] * p1y + F[8]);
      // epipolar_line_2 = F^T * p2
      const float l2x = F[0] * p2x + (F[3] * p2y + F[6]);
      const float l2y = F[1] * p2x + (F[4] * p2y + F[7]);
      const float l2z = F[2] * p2x + (F[5] * p2y + F[8]);
      const float d1 = std::fabs(p2x * l1x + (p2y * l1y + l1z)) / std::sqrt(l1x * l1x + l1y * l1y);
      const float d2 = std::fabs(p1x * l2x + (p1y * l2y + l2z)) / std::sqrt(l2x * l2x + l2y * l2y);
      cost += static_cast<double>(d1 + d2);
      ++n_joints_used;
    }
  }
  return cost;
}

struct PersonHypothesis {  // S3D:153-159 (+ detIdx: the association index the build defines)
  std::vector<std::array<V3f, NKP>> keypoints_normalized;
  std::vector<std::array<M2f, NKP>> keypoints_cov_normalized;
  std::vector<const float *> cameraExtrinsics;  // 3x4 row-major
  std::vector<int> cameraIDs;
  std::vector<float> score;
  std::vector<int> detIdx;
};

// calcCost (S3D:335-390)
double calcCost(const PersonHypothesis &hyp, const std::array<V3f, NKP> &det_kps, int det_ID, const float *Fs, int n_cam,
                double max_epi, bool &veto) {
  double total_cost = 0.;
  int n_obs_used = 0;
  int n_obs_in_hyp = (int)hyp.cameraIDs.size();
  if (n_obs_in_hyp == 0) { veto = true; return MAX_COSTS; }
  veto = false;
  double tmp_veto = 0.0;
  const double tolerance = 1.0 - 1.0 / (2 * n_obs_in_hyp), veto_delta = 1.0 / n_obs_in_hyp;
  for (int obs = 0; obs < n_obs_in_hyp; ++obs) {
    const float *F = Fs + 9 * (size_t)get_fundamental_idx(hyp.cameraIDs[obs], det_ID, n_cam);
    int n_joints_used;
    double cost = pair_cost(F, hyp.keypoints_normalized[obs].data(), det_kps.data(), n_joints_used);
    if (n_joints_used > 0) {
      cost /= n_joints_used;
      total_cost += cost;
      ++n_obs_used;
      if (cost > max_epi && (hyp.score[obs] > 0.5f || n_obs_in_hyp == 1)) {
        tmp_veto += veto_delta;
      } else if (cost > 2 * max_epi && (hyp.score[obs] > 0.5f || n_obs_in_hyp == 1)) {  // unreachable, kept (S3D:374-376)
        tmp_veto += 1;
      }
    }
  }
  if (tmp_veto > tolerance) veto = true;
  if (n_obs_used > 0) return total_cost / n_obs_used;
  veto = true;
  return MAX_COSTS;
}

// calcReprojectionError (S3D:425-438)
double calcReprojectionError(const V3f &X, const std::vector<const float *> &cams, const std::vector<V3f> &pts) {
  double averageError = 0., norm = 0.;
  for (size_t i = 0; i < cams.size(); ++i) {
    const float *M = cams[i];
    // (M * X.homogeneous()): M.leftCols<3>() * X (a0+(a1+a2)) + M.col(3)
    const float hx = (M[0] * X.x + (M[1] * X.y + M[2] * X.z)) + M[3];
    const float hy = (M[4] * X.x + (M[5] * X.y + M[6] * X.z)) + M[7];
    const float hz = (M[8] * X.x + (M[9] * X.y + M[10] * X.z)) + M[11];
    const float ix = hx / hz, iy = hy / hz;
    const float dx = ix - pts[i].x, dy = iy - pts[i].y;
    const float error = std::sqrt(dx * dx + dy * dy);
    averageError += static_cast<double>(pts[i].z * error);
    norm += static_cast<double>(pts[i].z);
  }
  return averageError / norm;
}

// triangulate (S3D:440-465)
V3f triangulate(const std::vector<const float *> &cams, const std::vector<V3f> &pts, bool weight_by_conf = false,
                double *reproj_error = nullptr) {
  const int n = (int)cams.size();
  const int rows = 2 * n;
  std::vector<float> A((size_t)rows * 4, 0.f);  // column-major
  for (int i = 0; i < n; ++i) {
    const float *P = cams[i];
    float r0[4], r1[4];
    for (int k = 0; k < 4; ++k) {
      r0[k] = pts[i].x * P[8 + k] - P[k];
      r1[k] = pts[i].y * P[8 + k] - P[4 + k];
    }
    float z0 = (r0[0] * r0[0] + r0[1] * r0[1]) + (r0[2] * r0[2] + r0[3] * r0[3]);
    if (z0 > 0.f) { float nn = std::sqrt(z0); for (int k = 0; k < 4; ++k) r0[k] = r0[k] / nn; }
    float z1 = (r1[0] * r1[0] + r1[1] * r1[1]) + (r1[2] * r1[2] + r1[3] * r1[3]);
    if (z1 > 0.f) { float nn = std::sqrt(z1); for (int k = 0; k < 4; ++k) r1[k] = r1[k] / nn; }
    if (weight_by_conf)
      for (int k = 0; k < 4; ++k) { r0[k] *= pts[i].z; r1[k] *= pts[i].z; }
    for (int k = 0; k < 4; ++k) { A[2 * i + (size_t)rows * k] = r0[k]; A[2 * i + 1 + (size_t)rows * k] = r1[k]; }
  }
  float v[4];
  null_vector_f32(rows, A, v);
  V3f X{v[0] / v[3], v[1] / v[3], v[2] / v[3]};  // hnormalized
  if (reproj_error) *reproj_error = calcReprojectionError(X, cams, pts);
  return X;
}

// mod_samples / draw_sigma_points / calc_covariance (S3D:471-523)
void calc_covariance(float cov[9], const V3f &mean_sample, const std::vector<V3f> &pts, const std::vector<M2f> &covs,
                     const std::vector<const float *> &cams) {
  const int num_cameras = (int)covs.size();
  const int dim = 2 * num_cameras;
  constexpr float kappa = 0.5f;
  const int num_samples = 2 * dim + 1;
  std::vector<float> weights(num_samples, 1.0f);
  weights[0] = 2 * kappa;
  for (float &w : weights) w /= (2.0f * (dim + kappa));
  std::vector<std::vector<V3f>> sigma_points(num_samples, pts);
  const float b = std::sqrt(dim + kappa);
  bool not_spd = false;
  for (int c = 0; c < num_cameras; ++c) {
    const M2f &A = covs[c];
    const float l11 = std::sqrt(A.a00);
    const float l21 = A.a10 / l11;
    const float l22 = std::sqrt(A.a11 - l21 * l21);
    if (!(std::isfinite(l11) && std::isfinite(l21) && std::isfinite(l22))) not_spd = true;
    const float dx1 = l11 * b, dy1 = l21 * b, dy2 = l22 * b;
    sigma_points[4 * c + 1][c].x -= dx1;
    sigma_points[4 * c + 1][c].y -= dy1;
    sigma_points[4 * c + 2][c].y -= dy2;
    sigma_points[4 * c + 3][c].x += dx1;
    sigma_points[4 * c + 3][c].y += dy1;
    sigma_points[4 * c + 4][c].y += dy2;
  }
  if (not_spd) {
    // DEFINED POLICY (the reference is undefined here): mod_samples takes square roots of the 2-D covariance entries
    // (S3D:473-475), so a view whose keypoint covariance is not SPD yields NaN sigma points, and what Eigen's JacobiSVD returns for a NaN
    // matrix is unspecified.  This library defines: the joint keeps its position and score, all entries of its 3-D covariance are NaN.
    for (int i = 0; i < 9; ++i) cov[i] = std::numeric_limits<float>::quiet_NaN();
    return;
  }
  std::vector<V3f> centered(num_samples);
  for (int s = 0; s < num_samples; ++s) {
    V3f X = triangulate(cams, sigma_points[s]);
    centered[s] = V3f{X.x - mean_sample.x, X.y - mean_sample.y, X.z - mean_sample.z};
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      float acc = 0.f;
      for (int s = 0; s < num_samples; ++s) {
        const float ci = (i == 0 ? centered[s].x : i == 1 ? centered[s].y : centered[s].z);
        const float cj = (j == 0 ? centered[s].x : j == 1 ? centered[s].y : centered[s].z);
        acc += (ci * weights[s]) * cj;
      }
      cov[i * 3 + j] = acc;
    }
}

inline double calcJointDist(const double a[3], const double b[3]) {  // S3D:467-469
  return std::sqrt((a[0] - b[0]) * (a[0] - b[0]) + (a[1] - b[1]) * (a[1] - b[1]) + (a[2] - b[2]) * (a[2] - b[2]));
}
inline void addToKeypointCovariance(se3d_keypoint3d &kp, double sigma) {  // S3D:273-277
  kp.cov[0] += sigma * sigma;
  kp.cov[3] += sigma * sigma;
  kp.cov[5] += sigma * sigma;
}

double calc_3D_dist(const se3d_person3d &p1, const se3d_person3d &p2) {  // S3D:392-408
  int n = 0;
  double dist = 0;
  for (int k = 0; k < NFU; ++k) {
    const auto &a = p1.keypoints[k], &b = p2.keypoints[k];
    if (a.score > 0 && b.score > 0) {
      dist += std::sqrt(std::pow(a.joint[0] - b.joint[0], 2) + std::pow(a.joint[1] - b.joint[1], 2) + std::pow(a.joint[2] - b.joint[2], 2));
      ++n;
    }
  }
  return n > 0 ? dist / n : MAX_COSTS;
}

void merge_persons(se3d_person3d &p1, const se3d_person3d &p2) {  // S3D:410-423
  for (int k = 0; k < NFU; ++k) {
    auto &kp1 = p1.keypoints[k];
    const auto &kp2 = p2.keypoints[k];
    double total = static_cast<double>(kp1.score + kp2.score);
    if (total > 0.0) {
      for (int a = 0; a < 3; ++a) kp1.joint[a] = ((double)kp1.score * kp1.joint[a] + (double)kp2.score * kp2.joint[a]) / total;
      kp1.score = std::max(kp1.score, kp2.score);
      for (int a = 0; a < 6; ++a) kp1.cov[a] = (kp1.cov[a] + kp2.cov[a]) / 2.0;
    }
  }
}

// per-hypothesis body of the OpenMP loop, first half (S3D:681-857): per-joint triangulation, outlier rejection, unscented covariance,
// scatter into the 21-slot fusion layout.  Returns the number of triangulated joints, -1 if the hypothesis has < 2 views.
int triangulate_joints(const PersonHypothesis &hyp, const ModelTables &T, const float *Fs, int n_cam, se3d_person3d &person_3d) {
  std::memset(&person_3d, 0, sizeof(person_3d));
  const int n_obs = (int)hyp.cameraIDs.size();
  if (n_obs < 2) return -1;
  int num_valid_keypoints = 0;
  for (int kp_idx = 0; kp_idx < NKP; ++kp_idx) {
    std::vector<V3f> jointOnEachCamera;
    std::vector<M2f> covarianceOnEachCamera;
    std::vector<const float *> cameraMatrices;
    std::vector<int> cameraIdx_joint;
    float avg_score = 0;
    for (int obs = 0; obs < n_obs; ++obs) {
      if (hyp.keypoints_normalized[obs][kp_idx].z >= g_triangulation_threshold) {
        jointOnEachCamera.push_back(hyp.keypoints_normalized[obs][kp_idx]);
        covarianceOnEachCamera.push_back(hyp.keypoints_cov_normalized[obs][kp_idx]);
        cameraMatrices.push_back(hyp.cameraExtrinsics[obs]);
        cameraIdx_joint.push_back(hyp.cameraIDs[obs]);
        avg_score += hyp.keypoints_normalized[obs][kp_idx].z;
      }
    }
    int numCameras = (int)cameraMatrices.size();
    if (numCameras < 2) continue;
    avg_score /= numCameras;

    double reprojError;
    V3f reconstructedPoint = triangulate(cameraMatrices, jointOnEachCamera, true, &reprojError);

    if (reprojError > g_reproj_error_max_acceptable && numCameras == 3) {  // S3D:748-792
      int bestIndex = -1;
      float bestDist = static_cast<float>(reprojError * reprojError);
      for (int i = 0; i < numCameras; ++i) {
        auto ptsSubset = jointOnEachCamera;
        auto camIdxSubset = cameraIdx_joint;
        ptsSubset.erase(ptsSubset.begin() + i);
        camIdxSubset.erase(camIdxSubset.begin() + i);
        const float p1x = ptsSubset[0].x, p1y = ptsSubset[0].y, p2x = ptsSubset[1].x, p2y = ptsSubset[1].y;
        const float *F = Fs + 9 * (size_t)get_fundamental_idx(camIdxSubset[0], camIdxSubset[1], n_cam);
        const float l1x = F[0] * p1x + (F[1] * p1y + F[2]);
        const float l1y = F[3] * p1x + (F[4] * p1y + F[5]);
        const float l1z = F[6] * p1x + (F[7] * p1y + F[8]);
        const float l2x = F[0] * p2x + (F[3] * p2y + F[6]);
        const float l2y = F[1] * p2x + (F[4] * p2y + F[7]);
        const float l2z = F[2] * p2x + (F[5] * p2y + F[8]);
        const float numerator_1 = p2x * l1x + (p2y * l1y + l1z);
        const float numerator_2 = p1x * l2x + (p1y * l2y + l2z);
        const float squared_symm_dist = numerator_1 * numerator_1 / (l1x * l1x + l1y * l1y) + numerator_2 * numerator_2 / (l2x * l2x + l2y * l2y);
        if (squared_symm_dist < bestDist) { bestDist = squared_symm_dist; bestIndex = i; }
      }
      if (bestIndex != -1) {
        stat(SE3D_ORACLE_STAT_OUTLIER3_REMOVED);
        cameraMatrices.erase(cameraMatrices.begin() + bestIndex);
        jointOnEachCamera.erase(jointOnEachCamera.begin() + bestIndex);
        covarianceOnEachCamera.erase(covarianceOnEachCamera.begin() + bestIndex);
        cameraIdx_joint.erase(cameraIdx_joint.begin() + bestIndex);
        reconstructedPoint = triangulate(cameraMatrices, jointOnEachCamera, true, &reprojError);
        avg_score = (jointOnEachCamera[0].z + jointOnEachCamera[1].z) / 2.0f;
      }
    } else if (reprojError > g_reproj_error_max_acceptable && numCameras >= 4) {  // S3D:793-838
      double bestReprojection = reprojError;
      int bestReprojectionIndex = -1;
      float bestScore = avg_score;
      V3f bestReconstructedPoint{0, 0, 0};
      for (int i = 0; i < numCameras; ++i) {
        auto camsSubset = cameraMatrices;
        auto ptsSubset = jointOnEachCamera;
        camsSubset.erase(camsSubset.begin() + i);
        ptsSubset.erase(ptsSubset.begin() + i);
        double projectionErrorSubset;
        V3f ptSubset = triangulate(camsSubset, ptsSubset, true, &projectionErrorSubset);
        if (bestReprojection > projectionErrorSubset && projectionErrorSubset < 0.9 * reprojError) {
          bestReprojection = projectionErrorSubset;
          bestReprojectionIndex = i;
          bestReconstructedPoint = ptSubset;
          float tmpScore = 0.;
          for (const auto &pt : ptsSubset) tmpScore += pt.z;
          bestScore = tmpScore / ptsSubset.size();
        }
      }
      if (bestReprojectionIndex != -1) {
        stat(SE3D_ORACLE_STAT_OUTLIER_LOO_REMOVED);
        cameraMatrices.erase(cameraMatrices.begin() + bestReprojectionIndex);
        jointOnEachCamera.erase(jointOnEachCamera.begin() + bestReprojectionIndex);
        covarianceOnEachCamera.erase(covarianceOnEachCamera.begin() + bestReprojectionIndex);
        cameraIdx_joint.erase(cameraIdx_joint.begin() + bestReprojectionIndex);
        reconstructedPoint = bestReconstructedPoint;
        reprojError = bestReprojection;
        avg_score = bestScore;
      }
    }
    if (reprojError > g_reproj_error_max_acceptable) { avg_score *= (g_reproj_error_max_acceptable / reprojError); stat(SE3D_ORACLE_STAT_SCORE_DOWNWEIGHTED); }  // S3D:840-844

    float covariance[9];
    calc_covariance(covariance, reconstructedPoint, jointOnEachCamera, covarianceOnEachCamera, cameraMatrices);

    se3d_keypoint3d &kp = person_3d.keypoints[T.kp2fusion[kp_idx]];
    kp.joint[0] = static_cast<double>(reconstructedPoint.x);
    kp.joint[1] = static_cast<double>(reconstructedPoint.y);
    kp.joint[2] = static_cast<double>(reconstructedPoint.z);
    kp.score = avg_score;
    kp.cov[0] = covariance[0]; kp.cov[1] = covariance[1]; kp.cov[2] = covariance[2];  // setKeypointCovariance (S3D:255-262)
    kp.cov[3] = covariance[4]; kp.cov[4] = covariance[5]; kp.cov[5] = covariance[8];
    if (covariance[0] != covariance[0]) stat(SE3D_ORACLE_STAT_NAN_COVARIANCE);  // non-SPD 2-D keypoint covariance: sqrt of a negative (S3D:473-475)
    ++num_valid_keypoints;
  }
  return num_valid_keypoints;
}

// second half (S3D:861-973): plausibility post-processing.  Returns true if the person is published.
bool post_filter(se3d_person3d &person_3d, const ModelTables &T, int num_valid_keypoints) {
  for (int kp_idx = 0; kp_idx < NKP; ++kp_idx) {  // limb-length covariance inflation (S3D:861-883)
    auto &kp = person_3d.keypoints[T.kp2fusion[kp_idx]];
    if (kp.score <= 0) continue;
    int parent_idx = T.kpParent[kp_idx];
    if (parent_idx >= 0) {
      const auto &parent_kp = person_3d.keypoints[T.kp2fusion[parent_idx]];
      if (parent_kp.score > 0 && T.limbLength[kp_idx] > 0) {
        double joint_dist = calcJointDist(kp.joint, parent_kp.joint);
        addToKeypointCovariance(kp, g_limbLCovOffsetSigma * (joint_dist - T.limbLength[kp_idx]) / T.limbLSigma[kp_idx]);
      } else if (T.simple && kp_idx == SIMPLE_RShoulder) {
        auto &kpLSh = person_3d.keypoints[T.kp2fusion[SIMPLE_LShoulder]];
        if (kpLSh.score > 0) {
          double joint_dist = calcJointDist(kp.joint, kpLSh.joint);
          addToKeypointCovariance(kp, g_limbLCovOffsetSigma * (joint_dist - simple_shoulderDist) / simple_shoulderSigma);
          addToKeypointCovariance(kpLSh, g_limbLCovOffsetSigma * (joint_dist - simple_shoulderDist) / simple_shoulderSigma);
        }
      }
    }
  }

  se3d_keypoint3d root;  // S3D:924-935
  std::memset(&root, 0, sizeof(root));
  if (person_3d.keypoints[F_MidHip].score > 0) root = person_3d.keypoints[F_MidHip];
  else if (person_3d.keypoints[F_LHip].score > 0 && person_3d.keypoints[F_RHip].score > 0) {
    for (int a = 0; a < 3; ++a) root.joint[a] = (person_3d.keypoints[F_LHip].joint[a] + person_3d.keypoints[F_RHip].joint[a]) / 2.;
    root.score = (person_3d.keypoints[F_LHip].score + person_3d.keypoints[F_RHip].score) / 2.f;
  }
  if (root.score > 0) {  // S3D:937-953
    for (int k = 0; k < NFU; ++k) {
      auto &kp = person_3d.keypoints[k];
      if (kp.score > 0) {
        double joint_dist = calcJointDist(root.joint, kp.joint);
        if (joint_dist > g_max_joint_dist_to_root) { std::memset(&kp, 0, sizeof(kp)); --num_valid_keypoints; stat(SE3D_ORACLE_STAT_ROOT_DIST_RESET); }
      } else {
        std::memset(&kp, 0, sizeof(kp));
        --num_valid_keypoints;
      }
    }
  }
  double feet_height = 0.0;  // S3D:955-966
  if (person_3d.keypoints[F_LAnkle].score > 0 && person_3d.keypoints[F_RAnkle].score > 0)
    feet_height = (person_3d.keypoints[F_LAnkle].joint[2] + person_3d.keypoints[F_RAnkle].joint[2]) / 2.0;
  else if (person_3d.keypoints[F_LAnkle].score > 0) feet_height = person_3d.keypoints[F_LAnkle].joint[2];
  else if (person_3d.keypoints[F_RAnkle].score > 0) feet_height = person_3d.keypoints[F_RAnkle].joint[2];
  if (std::abs(feet_height) > 0.50) { num_valid_keypoints = 0; stat(SE3D_ORACLE_STAT_FEET_REJECT); }
  if (!(num_valid_keypoints > g_min_num_valid_keypoints)) stat(SE3D_ORACLE_STAT_PERSON_REJECTED);
  return num_valid_keypoints > g_min_num_valid_keypoints;  // S3D:968
}

bool triangulate_hypothesis(const PersonHypothesis &hyp, const ModelTables &T, const float *Fs, int n_cam, se3d_person3d &person_3d) {
  const int n = triangulate_joints(hyp, T, Fs, n_cam, person_3d);
  if (n < 0) return false;
  return post_filter(person_3d, T, n);
}

// ------------------------------------------------------------------------------------------------------------
// stage 2 (pose_prior): GTSAM 4.0.3 semantics restated
// ------------------------------------------------------------------------------------------------------------
struct V3d {
  double x, y, z;
  V3d operator+(const V3d &o) const { return {x + o.x, y + o.y, z + o.z}; }
  V3d operator-(const V3d &o) const { return {x - o.x, y - o.y, z - o.z}; }
  V3d operator*(double s) const { return {x * s, y * s, z * s}; }
  V3d operator/(double s) const { return {x / s, y / s, z / s}; }
};
typedef std::map<int, V3d> Values;  // gtsam::Values keyed by fusion joint index (ascending iteration)

struct UnaryF { int key; double m[3]; double R[9]; };  // whitened error = R (q - m); R upper triangular (or diagonal)
struct RangeF { int k1, k2; double L, sigma; };
struct Graph { std::vector<UnaryF> unary; std::vector<RangeF> range; };

bool chol_lower(int n, std::vector<double> &A) {  // in-place LL^T, row-major full storage; false if not PD
  for (int j = 0; j < n; ++j) {
    double d = A[j * n + j];
    for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    A[j * n + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = A[i * n + j];
      for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = s / d;
    }
  }
  return true;
}
void chol_solve(int n, const std::vector<double> &L, std::vector<double> &b) {
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[i * n + k] * b[k];
    b[i] = s / L[i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = b[i];
    for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * b[k];
    b[i] = s / L[i * n + i];
  }
}

// noiseModel::Gaussian::Covariance(cov) (smart=true): diagonal covariance (|off-diag| <= 1e-9, upper triangle)
// -> Diagonal sigmas; else R = chol_upper(cov^-1) with R^T R = cov^-1.
bool make_whitener(const double cov[9], double R[9]) {
  for (int i = 0; i < 9; ++i) R[i] = 0.0;
  bool full = false;
  for (int i = 0; i < 3 && !full; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (std::fabs(cov[i * 3 + j]) > 1e-9) { full = true; break; }
  if (!full) {
    for (int i = 0; i < 3; ++i) R[i * 3 + i] = 1.0 / std::sqrt(cov[i * 3 + i]);
    return true;
  }
  // inverse (3x3 cofactors)
  const double a = cov[0], b = cov[1], c = cov[2], d = cov[3], e = cov[4], f = cov[5], g = cov[6], h = cov[7], i9 = cov[8];
  const double c00 = e * i9 - f * h, c01 = -(d * i9 - f * g), c02 = d * h - e * g;
  const double det = a * c00 + b * c01 + c * c02;
  const double id = 1.0 / det;
  double inv[9] = {c00 * id, -(b * i9 - c * h) * id, (b * f - c * e) * id,
                   c01 * id, (a * i9 - c * g) * id, -(a * f - c * d) * id,
                   c02 * id, -(a * h - b * g) * id, (a * e - b * d) * id};
  // upper Cholesky: inv = R^T R  (R = L^T of the lower factor)
  std::vector<double> L(inv, inv + 9);
  for (int r = 0; r < 3; ++r)
    for (int cc = r + 1; cc < 3; ++cc) L[r * 3 + cc] = L[cc * 3 + r] = 0.5 * (inv[r * 3 + cc] + inv[cc * 3 + r]);
  if (!chol_lower(3, L)) return false;
  for (int r = 0; r < 3; ++r)
    for (int cc = r; cc < 3; ++cc) R[r * 3 + cc] = L[cc * 3 + r];
  return true;
}

struct Linear {          // whitened Jacobian rows in dense form over the variable ordering `keys`
  int n = 0, m = 0;      // unknowns, rows
  std::vector<double> J; // [m][n]
  std::vector<double> b; // [m] = -whitened error
};

double graph_error(const Graph &g, const Values &v) {  // 0.5 * sum ||whitened error||^2
  double e = 0.0;
  for (const auto &u : g.unary) {
    const V3d &q = v.at(u.key);
    const double r[3] = {q.x - u.m[0], q.y - u.m[1], q.z - u.m[2]};
    for (int i = 0; i < 3; ++i) {
      double w = u.R[i * 3 + 0] * r[0] + u.R[i * 3 + 1] * r[1] + u.R[i * 3 + 2] * r[2];
      e += 0.5 * w * w;
    }
  }
  for (const auto &rf : g.range) {
    const V3d d = v.at(rf.k2) - v.at(rf.k1);
    const double r = std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
    const double w = (r - rf.L) / rf.sigma;
    e += 0.5 * w * w;
  }
  return e;
}

void linearize(const Graph &g, const Values &v, const std::map<int, int> &col, Linear &lin) {
  lin.n = 3 * (int)col.size();
  lin.m = 3 * (int)g.unary.size() + (int)g.range.size();
  lin.J.assign((size_t)lin.m * lin.n, 0.0);
  lin.b.assign(lin.m, 0.0);
  int row = 0;
  for (const auto &u : g.unary) {
    const V3d &q = v.at(u.key);
    const double r[3] = {q.x - u.m[0], q.y - u.m[1], q.z - u.m[2]};
    const int c0 = 3 * col.at(u.key);
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 3; ++j) lin.J[(size_t)(row + i) * lin.n + c0 + j] = u.R[i * 3 + j];  // R * I
      lin.b[row + i] = -(u.R[i * 3 + 0] * r[0] + u.R[i * 3 + 1] * r[1] + u.R[i * 3 + 2] * r[2]);
    }
    row += 3;
  }
  for (const auto &rf : g.range) {  // RangeFactor<Point3>: distance3 + norm3 Jacobian (H = d/r, or (1,1,1) if r <= 1e-10)
    const V3d d = v.at(rf.k2) - v.at(rf.k1);
    const double r = std::sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
    double H[3];
    if (std::fabs(r) > 1e-10) { H[0] = d.x / r; H[1] = d.y / r; H[2] = d.z / r; }
    else { H[0] = H[1] = H[2] = 1.0; }
    const int c1 = 3 * col.at(rf.k1), c2 = 3 * col.at(rf.k2);
    for (int j = 0; j < 3; ++j) {
      lin.J[(size_t)row * lin.n + c1 + j] = -H[j] / rf.sigma;
      lin.J[(size_t)row * lin.n + c2 + j] = H[j] / rf.sigma;
    }
    lin.b[row] = -(r - rf.L) / rf.sigma;
    row += 1;
  }
}

double linear_error(const Linear &lin, const std::vector<double> &delta) {  // 0.5 ||J d - b||^2
  double e = 0.0;
  for (int i = 0; i < lin.m; ++i) {
    double s = -lin.b[i];
    for (int j = 0; j < lin.n; ++j) s += lin.J[(size_t)i * lin.n + j] * delta[j];
    e += 0.5 * s * s;
  }
  return e;
}

// LevenbergMarquardtOptimizer(graph, init).optimize() with GTSAM 4.0.3 default parameters.
Values lm_optimize(const Graph &g, const Values &init, int *iterations_out, double *error_out) {
  const double lambdaInitial = 1e-5, lambdaFactor = 10.0, lambdaUpperBound = 1e5, lambdaLowerBound = 0.0;
  const double minModelFidelity = 1e-3, relativeErrorTol = 1e-5, absoluteErrorTol = 1e-5, errorTol = 0.0;
  const int maxIterations = 100;
  std::map<int, int> col;
  for (const auto &kv : init) { int c = (int)col.size(); col[kv.first] = c; }
  Values values = init;
  double error = graph_error(g, values);
  double lambda = lambdaInitial;
  int iterations = 0;
  auto finish = [&]() { if (iterations_out) *iterations_out = iterations; if (error_out) *error_out = error; return values; };
  if (error <= errorTol) return finish();  // NonlinearOptimizer::defaultOptimize
  double currentError;
  do {
    currentError = error;
    // ---- iterate(): linearise once, then tryLambda until it returns true
    Linear lin;
    linearize(g, values, col, lin);
    const int n = lin.n;
    std::vector<double> JtJ((size_t)n * n, 0.0), Jtb(n, 0.0);
    for (int i = 0; i < lin.m; ++i) {
      const double *Ji = &lin.J[(size_t)i * n];
      for (int a = 0; a < n; ++a) {
        if (Ji[a] == 0.0) continue;
        Jtb[a] += Ji[a] * lin.b[i];
        for (int c = 0; c < n; ++c) JtJ[(size_t)a * n + c] += Ji[a] * Ji[c];
      }
    }
    for (;;) {  // tryLambda
      std::vector<double> H = JtJ, delta = Jtb;
      for (int a = 0; a < n; ++a) H[(size_t)a * n + a] += lambda;  // sqrt(lambda)*I prior per variable
      bool solved = chol_lower(n, H);
      bool step_is_successful = false, stopSearchingLambda = false;
      double newError = std::numeric_limits<double>::infinity(), modelFidelity = 0.0;
      Values newValues;
      if (solved) {
        chol_solve(n, H, delta);
        const double newlinearizedError = linear_error(lin, delta);
        const double linearizedCostChange = error - newlinearizedError;
        if (linearizedCostChange >= 0) {
          newValues = values;
          for (auto &kv : newValues) {  // retract
            const int c = 3 * col.at(kv.first);
            kv.second = kv.second + V3d{delta[c], delta[c + 1], delta[c + 2]};
          }
          newError = graph_error(g, newValues);
          const double costChange = error - newError;
          if (linearizedCostChange > 1e-20) {
            modelFidelity = costChange / linearizedCostChange;
            step_is_successful = modelFidelity > minModelFidelity;
          }
          const double minAbsoluteTolerance = relativeErrorTol * error;
          if (std::fabs(costChange) < minAbsoluteTolerance) stopSearchingLambda = true;
        }
      }
      if (step_is_successful) {  // decreaseLambda (useFixedLambdaFactor)
        values = newValues;
        error = newError;
        lambda /= lambdaFactor;
        lambda = std::max(lambdaLowerBound, lambda);
        ++iterations;
        break;
      } else if (!stopSearchingLambda) {  // increaseLambda
        lambda *= lambdaFactor;
        if (lambda >= lambdaUpperBound) break;
      } else {
        break;
      }
    }
    // checkConvergence
    if (iterations >= maxIterations) break;
    if (error <= errorTol) break;
    const double absoluteDecrease = currentError - error;
    const double relativeDecrease = absoluteDecrease / currentError;
    const bool converged = (relativeErrorTol != 0.0 && relativeDecrease <= relativeErrorTol) || (absoluteDecrease <= absoluteErrorTol);
    if (converged) break;
  } while (std::isfinite(currentError));
  return finish();
}

// Marginals(graph, result) + marginalCovariance(key): 3x3 diagonal blocks of (J^T J)^-1 at the optimum.
bool marginal_covariances(const Graph &g, const Values &result, std::map<int, std::array<double, 9>> &out) {
  std::map<int, int> col;
  for (const auto &kv : result) { int c = (int)col.size(); col[kv.first] = c; }
  Linear lin;
  linearize(g, result, col, lin);
  const int n = lin.n;
  std::vector<double> H((size_t)n * n, 0.0);
  for (int i = 0; i < lin.m; ++i) {
    const double *Ji = &lin.J[(size_t)i * n];
    for (int a = 0; a < n; ++a) {
      if (Ji[a] == 0.0) continue;
      for (int c = 0; c < n; ++c) H[(size_t)a * n + c] += Ji[a] * Ji[c];
    }
  }
  if (!chol_lower(n, H)) return false;
  for (const auto &kv : col) {
    std::array<double, 9> blk;
    for (int j = 0; j < 3; ++j) {
      std::vector<double> e(n, 0.0);
      e[3 * kv.second + j] = 1.0;
      chol_solve(n, H, e);
      for (int i = 0; i < 3; ++i) blk[i * 3 + j] = e[3 * kv.second + i];
    }
    out[kv.first] = blk;
  }
  return true;
}

// addBinaryFactors (PP:384-481): {k1, k2, length, sigma, needs !Belly}
struct BoneDef { int k1, k2; double L, sigma; bool not_belly; };
const BoneDef bones_abs[] = {  // PP:434-479
    {8, 9, 0.134, 0.033, false},   {8, 12, 0.134, 0.033, false},  {9, 10, 0.449, 0.051, false},  {10, 11, 0.446, 0.051, false},
    {12, 13, 0.449, 0.051, false}, {13, 14, 0.446, 0.051, false}, {1, 0, 0.20, 0.025, false},    {1, 2, 0.15, 0.042, false},
    {1, 5, 0.15, 0.042, false},    {2, 3, 0.28, 0.045, false},    {3, 4, 0.25, 0.063, false},    {5, 6, 0.28, 0.045, false},
    {6, 7, 0.25, 0.063, false},    {8, 20, 0.23846, 0.071, false}, {20, 1, 0.25534, 0.035, false}, {0, 19, 0.11500, 0.035, false},
    {8, 1, 0.50, 0.071, true},     {0, 15, 0.05, 0.035, false},   {0, 16, 0.05, 0.035, false},   {15, 17, 0.10, 0.05, false},
    {16, 18, 0.10, 0.05, false}};
const BoneDef bones_norm[] = {  // PP:386-431
    {8, 9, 0.17, 0.062, false},    {8, 12, 0.17, 0.062, false},   {9, 10, 0.694, 0.111, false},  {10, 11, 0.708, 0.097, false},
    {12, 13, 0.694, 0.111, false}, {13, 14, 0.708, 0.097, false}, {1, 0, 0.33, 0.050, false},    {1, 2, 0.262, 0.092, false},
    {1, 5, 0.262, 0.092, false},   {2, 3, 0.515, 0.071, false},   {3, 4, 0.444, 0.084, false},   {5, 6, 0.515, 0.071, false},
    {6, 7, 0.444, 0.084, false},   {8, 20, 0.49, 0.05, false},    {20, 1, 0.51, 0.05, false},    {0, 19, 0.23, 0.05, false},
    {8, 1, 1.000, 0.02, true},     {0, 15, 0.085, 0.06, false},   {0, 16, 0.085, 0.06, false},   {15, 17, 0.167, 0.08, false},
    {16, 18, 0.167, 0.08, false}};

void addBinaryFactors(Graph &graph, const bool joints_measured[NFU], bool norm_height) {
  const BoneDef *bones = norm_height ? bones_norm : bones_abs;
  const double limbLSigmaFactor = norm_height ? 2.0 : 1.0;  // PP:934-937
  for (int i = 0; i < 21; ++i) {
    const BoneDef &b = bones[i];
    if (joints_measured[b.k1] && joints_measured[b.k2] && (!b.not_belly || !joints_measured[F_Belly]))
      graph.range.push_back(RangeF{b.k1, b.k2, b.L, b.sigma * limbLSigmaFactor});
  }
}

struct TrackingHypothesis {  // PP:68-121
  Values prevEstimate;
  std::array<std::array<V3d, 3>, NFU> velBuffer;
  double t_prev;
  int num_obs;
  int id;
  double height_prev;
  V3d root_prev;
  TrackingHypothesis(int id_, double t_create) : t_prev(t_create), num_obs(0), id(id_), height_prev(-1.0), root_prev{0, 0, 0} {
    for (auto &b : velBuffer) for (auto &v : b) v = V3d{0, 0, 0};
  }
  double calc_normed_dist(const se3d_person3d &person, double t) const {  // PP:84-101
    double delta_t = t - t_prev;
    int n = 0;
    double dist = 0;
    for (int k = 0; k < NFU; ++k) {
      const auto &kp = person.keypoints[k];
      auto it = prevEstimate.find(k);
      if (kp.score > 0.10f && it != prevEstimate.end()) {
        const V3d prev_kp = it->second * height_prev + root_prev;
        dist += std::sqrt(std::pow(kp.joint[0] - prev_kp.x, 2) + std::pow(kp.joint[1] - prev_kp.y, 2) + std::pow(kp.joint[2] - prev_kp.z, 2)) /
                (vel_sigmas[k] * delta_t);
        ++n;
      }
    }
    return n > 0 ? dist / n : 1e6;
  }
  double calc_3d_dist(const TrackingHypothesis &other) const {  // PP:103-119
    int n = 0;
    double dist = 0;
    for (const auto &kv : prevEstimate) {
      auto it = other.prevEstimate.find(kv.first);
      if (it != other.prevEstimate.end()) {
        const V3d kp = kv.second * height_prev + root_prev;
        const V3d okp = it->second * other.height_prev + other.root_prev;
        dist += std::sqrt(std::pow(kp.x - okp.x, 2) + std::pow(kp.y - okp.y, 2) + std::pow(kp.z - okp.z, 2));
        ++n;
      }
    }
    return n > 0 ? dist / n : 1e6;
  }
};

inline void cov6_to_mat(const double c[6], double M[9]) {
  M[0] = c[0]; M[1] = c[1]; M[2] = c[2]; M[3] = c[1]; M[4] = c[3]; M[5] = c[4]; M[6] = c[2]; M[7] = c[4]; M[8] = c[5];
}

}  // namespace

struct se3d_oracle_pp {  // global state of one pose_prior node (PP:54-60,123)
  se3d_params params;
  std::vector<TrackingHypothesis> tracks;
  double fb_delay_buffer[3];
  double t_prev;
  int next_id;
  int frame_nr;
  // bookkeeping of the last step for the parity tests (not part of the reference's state): the track id every detection was associated with /
  // created (PP:570-580), and the (removed id, surviving id) pairs of the track merges (PP:870-903)
  std::vector<int> last_det_ids, last_merges;
};

// ============================================================================================================
// exported API
// ============================================================================================================
extern "C" {

int se3d_oracle_rig_from_transforms(int32_t n_cam, const double *T, float *Rt_out, float *F_out) {
  if (n_cam < 2 || !T) return SE3D_ERR_ARG;
  std::vector<std::array<double, 12>> Ps(n_cam);
  std::vector<std::array<double, 4>> Cs(n_cam);
  for (int i = 0; i < n_cam; ++i) {
    const double *M = T + 16 * (size_t)i;
    for (int k = 0; k < 12; ++k) Ps[i][k] = M[k];
    // C_i = T_i^-1 .col(3) for an isometry: -R^T t (homogeneous 1)
    for (int a = 0; a < 3; ++a) Cs[i][a] = -(M[0 * 4 + a] * M[3] + M[1 * 4 + a] * M[7] + M[2 * 4 + a] * M[11]);
    Cs[i][3] = 1.0;
    if (Rt_out)
      for (int k = 0; k < 12; ++k) Rt_out[12 * (size_t)i + k] = static_cast<float>(M[k]);
  }
  if (F_out) {
    size_t idx = 0;
    for (int i = 0; i < n_cam; ++i)
      for (int j = i + 1; j < n_cam; ++j, ++idx) {
        double e[3];
        for (int a = 0; a < 3; ++a) {
          e[a] = 0;
          for (int k = 0; k < 4; ++k) e[a] += Ps[j][a * 4 + k] * Cs[i][k];
        }
        const double ex[9] = {0, -e[2], e[1], e[2], 0, -e[0], -e[1], e[0], 0};  // cross_prod_matrix (S3D:230-234)
        double Pinv[12];
        pseudo_inv34d(Ps[i].data(), Pinv);
        // (e_ij_cross * Ps[j] * Pinv).cast<float>() (S3D:1201): operator* associates to the left -> (e_x P_j) P_i^+
        double exPj[12];
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 4; ++b) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += ex[a * 3 + k] * Ps[j][k * 4 + b];
            exPj[a * 4 + b] = s;
          }
        for (int a = 0; a < 3; ++a)
          for (int b = 0; b < 3; ++b) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += exPj[a * 4 + k] * Pinv[k * 3 + b];
            F_out[9 * idx + a * 3 + b] = static_cast<float>(s);
          }
      }
  }
  return SE3D_OK;
}

int se3d_oracle_hungarian(int32_t *assignment, double *cost, const double *dist, int32_t n_rows, int32_t n_cols) {
  if (!assignment || !cost || !dist) return SE3D_ERR_ARG;
  hungarian(assignment, cost, dist, n_rows, n_cols);
  return SE3D_OK;
}

int se3d_oracle_dlt(int32_t n, const float *Rt, const float *pts, int32_t weight_by_conf, float *X3, double *reproj) {
  if (n < 2) return SE3D_ERR_ARG;
  std::vector<const float *> cams(n);
  std::vector<V3f> p(n);
  for (int i = 0; i < n; ++i) { cams[i] = Rt + 12 * (size_t)i; p[i] = V3f{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]}; }
  double err;
  V3f X = triangulate(cams, p, weight_by_conf != 0, &err);
  X3[0] = X.x; X3[1] = X.y; X3[2] = X.z;
  if (reproj) *reproj = err;
  return SE3D_OK;
}

int se3d_oracle_ut_cov(int32_t n, const float *Rt, const float *pts, const float *cov2, const float *mean3, float *cov_out9) {
  if (n < 2) return SE3D_ERR_ARG;
  std::vector<const float *> cams(n);
  std::vector<V3f> p(n);
  std::vector<M2f> c(n);
  for (int i = 0; i < n; ++i) {
    cams[i] = Rt + 12 * (size_t)i;
    p[i] = V3f{pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]};
    c[i] = M2f{cov2[4 * i], cov2[4 * i + 1], cov2[4 * i + 2], cov2[4 * i + 3]};
  }
  calc_covariance(cov_out9, V3f{mean3[0], mean3[1], mean3[2]}, p, c, cams);
  return SE3D_OK;
}

double se3d_oracle_pair_cost(const float *F9, const float *kp_a, const float *kp_b, int32_t *n_joints_used) {
  V3f a[NKP], b[NKP];
  for (int k = 0; k < NKP; ++k) { a[k] = V3f{kp_a[3 * k], kp_a[3 * k + 1], kp_a[3 * k + 2]}; b[k] = V3f{kp_b[3 * k], kp_b[3 * k + 1], kp_b[3 * k + 2]}; }
  int n;
  double c = pair_cost(F9, a, b, n);
  if (n_joints_used) *n_joints_used = n;
  return n > 0 ? c / n : -1.0;
}

int se3d_oracle_triangulate(const se3d_rig *rig, const se3d_params *params, const se3d_frame_in *frame, se3d_frame_out *out,
                            int32_t omp_threads) {
  if (!rig || !params || !frame || !out) return SE3D_ERR_ARG;
  const int NUM_CAMERAS = rig->n_cam;
  if (frame->n_cam != NUM_CAMERAS) return SE3D_ERR_ARG;
  const ModelTables T = tables_for(params->pose_method);
  const double g_max_epipolar_error = params->max_epi_dist;
  out->n_persons = 0;
  out->n_hyp = 0;
  out->flags = 0;

  // S3D:538-555: cameras with at least one detection
  std::vector<int> cameraIDs;
  for (int i = 0; i < NUM_CAMERAS; ++i)
    if (frame->det_offset[i + 1] - frame->det_offset[i] > 0) cameraIDs.push_back(i);
  const int num_cams_with_det = (int)cameraIDs.size();
  if (num_cams_with_det < 2) { out->flags |= SE3D_FLAG_FEW_CAMERAS; return SE3D_OK; }  // S3D:557-560

  auto n_humans = [&](int ci) { return frame->det_offset[cameraIDs[ci] + 1] - frame->det_offset[cameraIDs[ci]]; };
  auto human = [&](int ci, int h) -> const se3d_person2d & { return frame->persons[frame->det_offset[cameraIDs[ci]] + h]; };

  std::vector<PersonHypothesis> H;
  int cam_idx = 0;
  while (H.size() == 0 && cam_idx < num_cams_with_det) {  // S3D:565-586
    const float *intr = rig->K + 4 * (size_t)cameraIDs[cam_idx];
    for (int human_idx = 0; human_idx < n_humans(cam_idx); ++human_idx) {
      PersonHypothesis hyp;
      hyp.cameraExtrinsics.push_back(rig->Rt + 12 * (size_t)cameraIDs[cam_idx]);
      hyp.cameraIDs.push_back(cameraIDs[cam_idx]);
      hyp.score.push_back(human(cam_idx, human_idx).score);
      hyp.detIdx.push_back(human_idx);
      std::array<V3f, NKP> kps;
      kps.fill(V3f{0.f, 0.f, -1.f});
      std::array<M2f, NKP> covs;
      covs.fill(M2f{0, 0, 0, 0});
      int num_valid = normalize_keypoints(kps, covs, intr, human(cam_idx, human_idx));
      if (num_valid > NKP / 2) {
        hyp.keypoints_normalized.push_back(kps);
        hyp.keypoints_cov_normalized.push_back(covs);
        H.push_back(hyp);
      }
    }
    ++cam_idx;
  }

  for (; cam_idx < num_cams_with_det; ++cam_idx) {  // S3D:588-674
    std::vector<std::array<V3f, NKP>> dets_kps;
    std::vector<std::array<M2f, NKP>> dets_covs;
    std::vector<float> dets_score;
    std::vector<int> dets_orig;
    const float *intr = rig->K + 4 * (size_t)cameraIDs[cam_idx];
    for (int det_idx = 0; det_idx < n_humans(cam_idx); ++det_idx) {
      std::array<V3f, NKP> kps;
      kps.fill(V3f{0.f, 0.f, -1.f});
      std::array<M2f, NKP> covs;
      covs.fill(M2f{0, 0, 0, 0});
      int num_valid = normalize_keypoints(kps, covs, intr, human(cam_idx, det_idx));
      if (num_valid > NKP / 2) {
        dets_kps.push_back(kps);
        dets_covs.push_back(covs);
        dets_score.push_back(human(cam_idx, det_idx).score);
        dets_orig.push_back(det_idx);
      }
    }
    const int n_hyp = (int)H.size();
    const int n_det = (int)dets_kps.size();
    if (n_det == 0) continue;

    std::vector<double> C((size_t)n_hyp * n_det);  // column-major, rows = hypotheses
    std::vector<int> assignment(n_hyp, -1);
    std::vector<char> mask((size_t)n_hyp * n_det, 0);
    // The cost of every (hypothesis, detection) pair first — the entries do not depend on one another, so with omp_threads > 1 they are
    // evaluated in parallel (test convenience for the 256-camera comparisons; the reference evaluates them inside the scan below, serially) —
    // then the reference's scan in its order: the LAST feasible detection of a hypothesis stays in `assignment` (S3D:616-626).
    std::vector<char> vetoed((size_t)n_hyp * n_det, 0);
    const long long n_items = (long long)n_hyp * n_det;
#ifdef _OPENMP
#pragma omp parallel for num_threads(std::max(1, std::min<int>(omp_threads, (int)std::min<long long>(n_items, 1 << 20)))) schedule(dynamic, 16) if (omp_threads > 1 && n_items > 64)
#endif
    for (long long it = 0; it < n_items; ++it) {
      const int det_idx = (int)(it / n_hyp), hyp_idx = (int)(it - (long long)det_idx * n_hyp);
      bool veto;
      C[hyp_idx + (size_t)n_hyp * det_idx] = calcCost(H[hyp_idx], dets_kps[det_idx], cameraIDs[cam_idx], rig->F, NUM_CAMERAS, g_max_epipolar_error, veto);
      vetoed[hyp_idx + (size_t)n_hyp * det_idx] = veto ? 1 : 0;
    }
    for (int det_idx = 0; det_idx < n_det; ++det_idx)
      for (int hyp_idx = 0; hyp_idx < n_hyp; ++hyp_idx) {
        const double c = C[hyp_idx + (size_t)n_hyp * det_idx];
        if (!vetoed[hyp_idx + (size_t)n_hyp * det_idx] && c < g_max_epipolar_error) {
          mask[hyp_idx + (size_t)n_hyp * det_idx] = 1;
          assignment[hyp_idx] = det_idx;
        }
      }
    bool ambiguous = false;  // S3D:628
    for (int d = 0; d < n_det && !ambiguous; ++d) {
      int cnt = 0;
      for (int h = 0; h < n_hyp; ++h) cnt += mask[h + (size_t)n_hyp * d];
      if (cnt > 1) ambiguous = true;
    }
    for (int h = 0; h < n_hyp && !ambiguous; ++h) {
      int cnt = 0;
      for (int d = 0; d < n_det; ++d) cnt += mask[h + (size_t)n_hyp * d];
      if (cnt > 1) ambiguous = true;
    }
    if (ambiguous) {
      double cost = 0.0;
      stat(SE3D_ORACLE_STAT_SWEEP_HUNGARIAN);
      hungarian(assignment.data(), &cost, C.data(), n_hyp, n_det);
    }
    auto new_hyp_from_det = [&](int det_idx) {
      PersonHypothesis hyp;
      hyp.cameraExtrinsics.push_back(rig->Rt + 12 * (size_t)cameraIDs[cam_idx]);
      hyp.cameraIDs.push_back(cameraIDs[cam_idx]);
      hyp.score.push_back(dets_score[det_idx]);
      hyp.detIdx.push_back(dets_orig[det_idx]);
      hyp.keypoints_normalized.push_back(dets_kps[det_idx]);
      hyp.keypoints_cov_normalized.push_back(dets_covs[det_idx]);
      H.push_back(hyp);
    };
    std::vector<char> det_handled(n_det, 0);
    for (int hyp_idx = 0; hyp_idx < n_hyp; ++hyp_idx) {
      int det_idx = assignment[hyp_idx];
      if (det_idx >= 0) {
        det_handled[det_idx] = 1;
        if (!mask[hyp_idx + (size_t)n_hyp * det_idx]) {
          stat(SE3D_ORACLE_STAT_SPAWN_UNMASKED);
          new_hyp_from_det(det_idx);
        } else {
          PersonHypothesis &h = H[hyp_idx];
          h.cameraExtrinsics.push_back(rig->Rt + 12 * (size_t)cameraIDs[cam_idx]);
          h.cameraIDs.push_back(cameraIDs[cam_idx]);
          h.score.push_back(dets_score[det_idx]);
          h.detIdx.push_back(dets_orig[det_idx]);
          h.keypoints_normalized.push_back(dets_kps[det_idx]);
          h.keypoints_cov_normalized.push_back(dets_covs[det_idx]);
        }
      }
    }
    for (int det_idx = 0; det_idx < n_det; ++det_idx)
      if (!det_handled[det_idx]) new_hyp_from_det(det_idx);
  }

  const int nH = (int)H.size();
  out->n_hyp = nH;
  if (out->hyp_assoc && out->hyp_capacity > 0) {
    for (int h = 0; h < std::min(nH, out->hyp_capacity); ++h) {
      int32_t *row = out->hyp_assoc + (size_t)h * NUM_CAMERAS;
      for (int c = 0; c < NUM_CAMERAS; ++c) row[c] = -1;
      for (size_t o = 0; o < H[h].cameraIDs.size(); ++o) row[H[h].cameraIDs[o]] = H[h].detIdx[o];
    }
    if (nH > out->hyp_capacity) out->flags |= SE3D_FLAG_HYP_OVERFLOW;
  }

  // S3D:676-982 — per-hypothesis triangulation; canonical (serial) order = ascending hyp_idx
  std::vector<se3d_person3d> persons(nH);
  std::vector<char> keep(nH, 0);
#ifdef _OPENMP
  if (omp_threads > 1 && nH > 0) {
#pragma omp parallel for num_threads(std::min<int>(omp_threads, nH)) schedule(dynamic, 1)
    for (int h = 0; h < nH; ++h) keep[h] = triangulate_hypothesis(H[h], T, rig->F, NUM_CAMERAS, persons[h]) ? 1 : 0;
  } else
#endif
  {
    (void)omp_threads;
    for (int h = 0; h < nH; ++h) keep[h] = triangulate_hypothesis(H[h], T, rig->F, NUM_CAMERAS, persons[h]) ? 1 : 0;
  }
  std::vector<se3d_person3d> list;
  std::vector<int> list_hyp;
  for (int h = 0; h < nH; ++h)
    if (keep[h]) { list.push_back(persons[h]); list_hyp.push_back(h); }

  for (size_t i = 0; i < list.size(); ++i)  // S3D:984-996
    for (size_t j = i + 1; j < list.size();) {
      if (calc_3D_dist(list[i], list[j]) < g_merge_dist_thresh) {
        stat(SE3D_ORACLE_STAT_DUPLICATE_MERGE);
        merge_persons(list[i], list[j]);
        list.erase(list.begin() + j);
        list_hyp.erase(list_hyp.begin() + j);
      } else
        ++j;
    }
  if ((int)list.size() > out->capacity) return SE3D_ERR_CAPACITY;
  out->n_persons = (int)list.size();
  for (size_t i = 0; i < list.size(); ++i) {
    out->persons[i] = list[i];
    if (out->person_hyp) out->person_hyp[i] = list_hyp[i];
  }
  return SE3D_OK;
}

// ---------------- stage 2 ----------------
// Config 2 of SURVEY 8d: rows N, T, P, O, U with the association GIVEN (no sweep, no post-processing, no merge).
int se3d_oracle_triangulate_assoc(const se3d_rig *rig, const se3d_params *params, const se3d_frame_in *frame, int32_t n_hyp,
                                  const int32_t *assoc, se3d_person3d *persons_out) {
  if (!rig || !params || !frame || !assoc || !persons_out || n_hyp < 0) return SE3D_ERR_ARG;
  const int N = rig->n_cam;
  if (frame->n_cam != N) return SE3D_ERR_ARG;
  const ModelTables T = tables_for(params->pose_method);
  for (int h = 0; h < n_hyp; ++h) {
    PersonHypothesis hyp;
    for (int c = 0; c < N; ++c) {  // ascending camera order = the order the sweep attaches views in (S3D:588)
      const int det = assoc[(size_t)h * N + c];
      if (det < 0) continue;
      if (det >= frame->det_offset[c + 1] - frame->det_offset[c]) return SE3D_ERR_ARG;
      const se3d_person2d &human = frame->persons[frame->det_offset[c] + det];
      std::array<V3f, NKP> kps;
      kps.fill(V3f{0.f, 0.f, -1.f});
      std::array<M2f, NKP> covs;
      covs.fill(M2f{0, 0, 0, 0});
      if (normalize_keypoints(kps, covs, rig->K + 4 * (size_t)c, human) <= NKP / 2) continue;  // the sweep never attaches such a detection (S3D:579, 599)
      hyp.cameraExtrinsics.push_back(rig->Rt + 12 * (size_t)c);
      hyp.cameraIDs.push_back(c);
      hyp.score.push_back(human.score);
      hyp.detIdx.push_back(det);
      hyp.keypoints_normalized.push_back(kps);
      hyp.keypoints_cov_normalized.push_back(covs);
    }
    triangulate_joints(hyp, T, rig->F, N, persons_out[h]);
  }
  return SE3D_OK;
}

void se3d_oracle_stats_reset(void) {
  for (auto &c : g_stat) c.store(0);
}
void se3d_oracle_stats_get(int64_t *out) {
  for (int i = 0; i < SE3D_ORACLE_STAT_COUNT; ++i) out[i] = (int64_t)g_stat[i].load();
}

// The restated node constants and per-pose tables as data (tests/golden/pose_constants.json holds the reference's values by name).
// scalars13: g_min_num_valid_keypoints, g_triangulation_threshold, g_reproj_error_max_acceptable, g_max_joint_dist_to_root, g_merge_dist_thresh,
//            g_limbLCovOffsetSigma, MAX_COSTS | g_min_score, g_predNoiseSigma, g_defaultResSigma, g_root_sigma_factor, g_n_mov_avg,
//            g_t_max_unobserved, g_dist_threshold, g_min_num_obs_track, g_avg_delay  (16 values)
void se3d_oracle_node_constants(double *scalars16) {
  const double v[16] = {(double)g_min_num_valid_keypoints, (double)g_triangulation_threshold, g_reproj_error_max_acceptable, g_max_joint_dist_to_root,
                        g_merge_dist_thresh, g_limbLCovOffsetSigma, MAX_COSTS, (double)g_min_score, g_predNoiseSigma, g_defaultResSigma,
                        g_root_sigma_factor, (double)g_n_mov_avg, g_t_max_unobserved, g_dist_threshold, (double)g_min_num_obs_track, g_avg_delay};
  for (int i = 0; i < 16; ++i) scalars16[i] = v[i];
}
void se3d_oracle_pose_tables(int32_t pose_method, int32_t *kp_parent17, double *limb_length17, double *limb_sigma17, int32_t *kp2fusion17,
                             double *shoulder2) {
  const auto T = tables_for(pose_method);
  for (int k = 0; k < NKP; ++k) { kp_parent17[k] = T.kpParent[k]; limb_length17[k] = T.limbLength[k]; limb_sigma17[k] = T.limbLSigma[k]; kp2fusion17[k] = T.kp2fusion[k]; }
  shoulder2[0] = simple_shoulderDist; shoulder2[1] = simple_shoulderSigma;
}

// The 21 range factors of addBinaryFactors (PP:384-481) per branch: k1, k2, length, sigma coefficient, "only if the belly is NOT measured".
void se3d_oracle_bone_table(int32_t norm_height, int32_t *k1, int32_t *k2, double *length, double *sigma, int32_t *not_belly) {
  const BoneDef *bones = norm_height ? bones_norm : bones_abs;
  for (int i = 0; i < 21; ++i) { k1[i] = bones[i].k1; k2[i] = bones[i].k2; length[i] = bones[i].L; sigma[i] = bones[i].sigma; not_belly[i] = bones[i].not_belly ? 1 : 0; }
}

// The restated FUSION_BODY_PARTS constants (FBP:3-24, 33) as data, so that a test can compare them with the reference's header compiled
// in place (oracle/_ref/libfbp_ref.so, tests/golden/fusion_body_parts.json).
void se3d_oracle_fusion_constants(int32_t *indices22, double *vel_sigmas21) {
  const int v[22] = {F_Nose, F_Neck, F_RShoulder, F_RElbow, F_RWrist, F_LShoulder, F_LElbow, F_LWrist, F_MidHip, F_RHip, F_RKnee, F_RAnkle,
                     F_LHip, F_LKnee, F_LAnkle, F_REye, F_LEye, F_REar, F_LEar, F_Head, F_Belly, NFU};
  for (int i = 0; i < 22; ++i) indices22[i] = v[i];
  for (int i = 0; i < NFU; ++i) vel_sigmas21[i] = vel_sigmas[i];
}

se3d_oracle_pp *se3d_oracle_pp_create(const se3d_params *params) {
  if (!params) return nullptr;
  se3d_oracle_pp *pp = new se3d_oracle_pp();
  pp->params = *params;
  for (double &d : pp->fb_delay_buffer) d = 0.10;  // g_avg_delay (PP:51,54)
  pp->t_prev = 0.0;
  pp->next_id = 0;
  pp->frame_nr = 0;
  return pp;
}
void se3d_oracle_pp_destroy(se3d_oracle_pp *pp) { delete pp; }

int se3d_oracle_pp_step(se3d_oracle_pp *pp, int32_t n_det, const se3d_person3d *dets, int32_t n_cam, const float *fb_delay_per_cam,
                        double t, se3d_fused_out *out, int32_t omp_threads) {
  if (!pp || !out || (n_det > 0 && !dets)) return SE3D_ERR_ARG;
  const bool simple = pp->params.pose_method == SE3D_POSE_SIMPLE;
  const bool norm_height = pp->params.norm_height != 0;
  auto &g_tracks = pp->tracks;
  out->n_persons = 0;
  pp->last_det_ids.clear();
  pp->last_merges.clear();

  double curr_avg_delay = 0.0;  // PP:513-526
  int n_valid_delay_meas = 0;
  for (int i = 0; i < n_cam; ++i) {
    const float delay = fb_delay_per_cam ? fb_delay_per_cam[i] : -1.0f;
    if (delay > 0.0f) { curr_avg_delay += static_cast<double>(delay); ++n_valid_delay_meas; }
  }
  if (n_valid_delay_meas > 0) curr_avg_delay /= n_valid_delay_meas;
  else curr_avg_delay = g_avg_delay;
  pp->fb_delay_buffer[pp->frame_nr % g_n_mov_avg] = curr_avg_delay;
  double pred_delta_t = (0.0 + pp->fb_delay_buffer[0] + pp->fb_delay_buffer[1] + pp->fb_delay_buffer[2]) / g_n_mov_avg;
  out->pred_delta_t = (float)pred_delta_t;

  auto remove_old_tracks = [&]() {  // PP:191-211
    for (const auto &tr : g_tracks) if (t - tr.t_prev > g_t_max_unobserved) stat(SE3D_ORACLE_STAT_TRACK_TIMEOUT);
    g_tracks.erase(std::remove_if(g_tracks.begin(), g_tracks.end(), [&](const TrackingHypothesis &tr) { return t - tr.t_prev > g_t_max_unobserved; }),
                   g_tracks.end());
  };
  const int n_hyp = (int)g_tracks.size();
  if (n_det == 0) {  // PP:537-546
    stat(SE3D_ORACLE_STAT_EMPTY_FRAME);
    remove_old_tracks();
    pp->t_prev = t;
    out->n_tracks = (int)g_tracks.size();
    return SE3D_OK;
  }

  std::vector<int> assignment;
  if (n_hyp > 0) {  // PP:550-568
    std::vector<double> C((size_t)n_det * n_hyp);
    assignment.assign(n_det, -1);
    double cost = 0.0;
    for (int tr = 0; tr < n_hyp; ++tr)
      for (int p = 0; p < n_det; ++p) C[p + (size_t)n_det * tr] = g_tracks[tr].calc_normed_dist(dets[p], t);
    hungarian(assignment.data(), &cost, C.data(), n_det, n_hyp);
    for (int i = 0; i < n_det; ++i)
      if (assignment[i] >= 0 && C[i + (size_t)n_det * assignment[i]] > g_dist_threshold) { assignment[i] = -1; stat(SE3D_ORACLE_STAT_TRACK_ASSOC_REJECTED); }
  }
  std::vector<int> track_ids(n_det);
  for (int p = 0; p < n_det; ++p) {  // PP:570-580
    if (!assignment.empty() && assignment[p] >= 0) track_ids[p] = assignment[p];
    else {
      g_tracks.push_back(TrackingHypothesis(pp->next_id, t));
      track_ids[p] = (int)g_tracks.size() - 1;
      ++pp->next_id;
    }
  }

  for (int p = 0; p < n_det; ++p) pp->last_det_ids.push_back(g_tracks[track_ids[p]].id);
  std::vector<se3d_person3d> fused(n_det), fused_pred(n_det);
  std::vector<char> publish(n_det, 0);
  std::vector<int> lm_iters(n_det, 0);
  std::vector<double> lm_obj(n_det, -1.0);   // graph.error(result) per detection (se3d_fused_out.lm_objective); -1: optimisation failed

  auto process_person = [&](int person_idx) {  // PP:587-853
    const se3d_person3d &person = dets[person_idx];
    TrackingHypothesis &curr_track = g_tracks[track_ids[person_idx]];
    se3d_person3d &person_fused = fused[person_idx];
    se3d_person3d &person_fused_pred = fused_pred[person_idx];
    std::memset(&person_fused, 0, sizeof(se3d_person3d));
    std::memset(&person_fused_pred, 0, sizeof(se3d_person3d));
    person_fused.id = curr_track.id;
    person_fused_pred.id = curr_track.id;

    int num_meas = 0;
    bool joints_measured[NFU] = {false}, use_velocity[NFU] = {false};
    Graph graph;
    Values measurements;
    se3d_keypoint3d root, neck;
    std::memset(&root, 0, sizeof(root));
    std::memset(&neck, 0, sizeof(neck));
    double height = 1.0;
    if (!simple) {
      root = person.keypoints[F_MidHip];
      neck = person.keypoints[F_Neck];
    } else {
      const auto &hl = person.keypoints[F_LHip], &hr = person.keypoints[F_RHip];
      if (hl.score > 0.0f && hr.score > 0.0f) {
        for (int a = 0; a < 3; ++a) root.joint[a] = (hl.joint[a] + hr.joint[a]) / 2.0;
        root.score = (hl.score + hr.score) / 2.0f;
      }
      const auto &sl = person.keypoints[F_LShoulder], &sr = person.keypoints[F_RShoulder];
      if (sl.score > 0.0f && sr.score > 0.0f) {
        for (int a = 0; a < 3; ++a) neck.joint[a] = (sl.joint[a] + sr.joint[a]) / 2.0;
        neck.score = (sl.score + sr.score) / 2.0f;
      }
    }
    bool whiten_ok = true;
    auto add_unary = [&](int key, double mx, double my, double mz, const double cov[9]) {
      UnaryF u;
      u.key = key; u.m[0] = mx; u.m[1] = my; u.m[2] = mz;
      if (!make_whitener(cov, u.R)) whiten_ok = false;
      graph.unary.push_back(u);
    };
    if (root.score > g_min_score) {  // PP:658-694
      if (norm_height) {
        if (neck.score > g_min_score)
          height = std::sqrt((neck.joint[0] - root.joint[0]) * (neck.joint[0] - root.joint[0]) + (neck.joint[1] - root.joint[1]) * (neck.joint[1] - root.joint[1]) +
                             (neck.joint[2] - root.joint[2]) * (neck.joint[2] - root.joint[2]));
        else height = 0.60;
      }
      double root_cov[9];
      if (!simple) cov6_to_mat(root.cov, root_cov);
      else {
        double cl[9], cr[9];
        cov6_to_mat(person.keypoints[F_LHip].cov, cl);
        cov6_to_mat(person.keypoints[F_RHip].cov, cr);
        for (int i = 0; i < 9; ++i) root_cov[i] = (cl[i] + cr[i]) / 2.0;
      }
      double c[9];
      for (int i = 0; i < 9; ++i) c[i] = root_cov[i] / (height * height) / (g_root_sigma_factor * g_root_sigma_factor);
      add_unary(F_MidHip, 0.0, 0.0, 0.0, c);
      measurements[F_MidHip] = V3d{0, 0, 0};
      ++num_meas;
      joints_measured[F_MidHip] = true;
    }
    if (curr_track.height_prev < 0.0) {  // PP:699-702
      curr_track.height_prev = height;
      curr_track.root_prev = V3d{root.joint[0], root.joint[1], root.joint[2]};
    }
    for (int k = 0; k < NFU; ++k) {  // PP:704-719
      if (k == F_MidHip) continue;
      // "simple": a measured Neck slot plus the derived neck would insert key 1 twice (gtsam::Values::insert throws, the
      // reference node would die); stage 1 never fills that slot in "simple" mode.  Defined here: the derived neck wins.
      if (simple && k == F_Neck && neck.score > g_min_score) continue;
      const auto &kp = person.keypoints[k];
      if (kp.score > g_min_score) {
        double cov[9], c[9];
        cov6_to_mat(kp.cov, cov);
        for (int i = 0; i < 9; ++i) c[i] = cov[i] / (height * height);
        const double mx = (kp.joint[0] - root.joint[0]) / height, my = (kp.joint[1] - root.joint[1]) / height, mz = (kp.joint[2] - root.joint[2]) / height;
        add_unary(k, mx, my, mz, c);
        measurements[k] = V3d{mx, my, mz};
        ++num_meas;
        joints_measured[k] = true;
      }
    }
    if (simple && neck.score > g_min_score) {  // PP:721-737
      double cl[9], cr[9], c[9];
      cov6_to_mat(person.keypoints[F_LShoulder].cov, cl);
      cov6_to_mat(person.keypoints[F_RShoulder].cov, cr);
      for (int i = 0; i < 9; ++i) c[i] = (cl[i] + cr[i]) / 2.0 / (height * height);
      const double mx = (neck.joint[0] - root.joint[0]) / height, my = (neck.joint[1] - root.joint[1]) / height, mz = (neck.joint[2] - root.joint[2]) / height;
      add_unary(F_Neck, mx, my, mz, c);
      measurements[F_Neck] = V3d{mx, my, mz};  // Values::insert on an existing key would throw in GTSAM; cannot happen in "simple" (slot 1 is never measured)
      ++num_meas;
      joints_measured[F_Neck] = true;
    }
    if (num_meas == 0) return;  // PP:739-741

    {  // setInitialState (PP:483-503)
      std::vector<int> keys_to_remove;
      for (const auto &kv : curr_track.prevEstimate)
        if (!measurements.count(kv.first)) keys_to_remove.push_back(kv.first);
      for (int key : keys_to_remove) {
        stat(SE3D_ORACLE_STAT_JOINT_VANISHED);
        curr_track.prevEstimate.erase(key);
        for (auto &v : curr_track.velBuffer[key]) v = V3d{0, 0, 0};
      }
      for (const auto &kv : measurements) {
        if (!curr_track.prevEstimate.count(kv.first)) curr_track.prevEstimate[kv.first] = kv.second;
        else use_velocity[kv.first] = true;
      }
    }
    addBinaryFactors(graph, joints_measured, norm_height);

    Values result;
    int iters = 0;
    double final_error = -1.0;
    if (whiten_ok) result = lm_optimize(graph, curr_track.prevEstimate, &iters, &final_error);
    else { result = measurements; stat(SE3D_ORACLE_STAT_LM_FAILED); }  // optimisation failed -> output measurements (PP:751-758)
    lm_iters[person_idx] = iters;
    lm_obj[person_idx] = final_error;

    std::map<int, std::array<double, 9>> marg;
    bool use_marginals = whiten_ok && marginal_covariances(graph, result, marg);  // PP:760-767
    if (!use_marginals) stat(SE3D_ORACLE_STAT_MARGINALS_FAILED);  // -> default covariance (PP:788-793)

    for (const auto &kv : result) {  // PP:770-837
      const int k = kv.first;
      double joint_fused[3] = {kv.second.x * height + root.joint[0], kv.second.y * height + root.joint[1], kv.second.z * height + root.joint[2]};
      for (int a = 0; a < 3; ++a) person_fused.keypoints[k].joint[a] = joint_fused[a];
      if (k == F_MidHip) person_fused.keypoints[k].score = std::max(g_min_score, root.score);
      else if (k == F_Neck) person_fused.keypoints[k].score = std::max(g_min_score, neck.score);
      else person_fused.keypoints[k].score = std::max(g_min_score, person.keypoints[k].score);
      double cov[9];
      if (use_marginals) {
        const auto &m = marg.at(k);
        for (int i = 0; i < 9; ++i) cov[i] = m[i] * height * height;
      } else {
        for (int i = 0; i < 9; ++i) cov[i] = 0.0;
        cov[0] = cov[4] = cov[8] = g_defaultResSigma * g_defaultResSigma;
      }
      if (k == F_MidHip)
        for (int i = 0; i < 9; ++i) cov[i] *= (g_root_sigma_factor * g_root_sigma_factor);
      auto &kc = person_fused.keypoints[k].cov;
      kc[0] = cov[0]; kc[1] = cov[1]; kc[2] = cov[2]; kc[3] = cov[4]; kc[4] = cov[5]; kc[5] = cov[8];

      double joint_pred[3] = {joint_fused[0], joint_fused[1], joint_fused[2]};
      if (use_velocity[k]) {  // PP:819-826
        const V3d cur = kv.second * height + V3d{root.joint[0], root.joint[1], root.joint[2]};
        const V3d prev = curr_track.prevEstimate.at(k) * curr_track.height_prev + curr_track.root_prev;
        curr_track.velBuffer[k][pp->frame_nr % g_n_mov_avg] = (cur - prev) / (t - pp->t_prev);
        V3d acc{0, 0, 0};
        for (const auto &v : curr_track.velBuffer[k]) acc = acc + v;
        const V3d pred_delta = acc / (double)g_n_mov_avg * pred_delta_t;
        joint_pred[0] += pred_delta.x; joint_pred[1] += pred_delta.y; joint_pred[2] += pred_delta.z;
      }
      auto &pk = person_fused_pred.keypoints[k];
      for (int a = 0; a < 3; ++a) pk.joint[a] = joint_pred[a];
      pk.score = person_fused.keypoints[k].score;
      for (int a = 0; a < 6; ++a) pk.cov[a] = kc[a];
      addToKeypointCovariance(pk, g_predNoiseSigma);
    }
    curr_track.t_prev = t;  // PP:839-843
    curr_track.prevEstimate.swap(result);
    curr_track.height_prev = height;
    curr_track.root_prev = V3d{root.joint[0], root.joint[1], root.joint[2]};
    ++curr_track.num_obs;
    if (curr_track.num_obs > g_min_num_obs_track) publish[person_idx] = 1;
  };
#ifdef _OPENMP
  if (omp_threads > 1) {
#pragma omp parallel for num_threads(std::min<int>(omp_threads, n_det)) schedule(dynamic, 1)
    for (int p = 0; p < n_det; ++p) process_person(p);
  } else
#endif
  {
    (void)omp_threads;
    for (int p = 0; p < n_det; ++p) process_person(p);
  }

  std::vector<int> out_idx;
  for (int p = 0; p < n_det; ++p)
    if (publish[p]) out_idx.push_back(p);
  if ((int)out_idx.size() > out->capacity) return SE3D_ERR_CAPACITY;
  out->n_persons = (int)out_idx.size();
  for (size_t i = 0; i < out_idx.size(); ++i) {
    out->fused[i] = fused[out_idx[i]];
    out->fused_pred[i] = fused_pred[out_idx[i]];
    if (out->lm_iterations) out->lm_iterations[i] = lm_iters[out_idx[i]];
    if (out->lm_objective) out->lm_objective[i] = lm_obj[out_idx[i]];
  }

  remove_old_tracks();  // PP:867
  for (size_t i = 0; i < g_tracks.size(); ++i)  // PP:870-903
    for (size_t j = i + 1; j < g_tracks.size();) {
      if (g_tracks[i].calc_3d_dist(g_tracks[j]) < g_merge_dist_thresh) {
        const int id_to_remove = g_tracks[j].id;
        pp->last_merges.push_back(id_to_remove);
        pp->last_merges.push_back(g_tracks[i].id);
        g_tracks.erase(g_tracks.begin() + j);
        stat(SE3D_ORACLE_STAT_TRACK_MERGE);
        for (int p = 0; p < out->n_persons; ++p)
          if ((int)out->fused[p].id == id_to_remove) { out->fused[p].id = g_tracks[i].id; out->fused_pred[p].id = g_tracks[i].id; stat(SE3D_ORACLE_STAT_ID_REWRITE); }
      } else
        ++j;
    }
  out->n_tracks = (int)g_tracks.size();
  pp->t_prev = t;
  ++pp->frame_nr;
  return SE3D_OK;
}

// Test bookkeeping of the last se3d_oracle_pp_step: ids[p] = id of the track detection p updated or created; pairs[2 m], pairs[2 m + 1] =
// the id removed by merge m and the id that survived.  Both return the number of entries (detections / merges) the step had.
int32_t se3d_oracle_pp_last_det_ids(const se3d_oracle_pp *pp, int32_t *ids, int32_t capacity) {
  if (!pp) return -1;
  for (size_t i = 0; i < pp->last_det_ids.size() && (int32_t)i < capacity; ++i) ids[i] = pp->last_det_ids[i];
  return (int32_t)pp->last_det_ids.size();
}
// live tracks after the last step, in list order (test bookkeeping): id_obs[2 i] = id, id_obs[2 i + 1] = num_obs; state[70 i ..] = t_prev,
// height_prev, root_prev[3], mask of the joints in prevEstimate (as a double), prevEstimate[21][3] (zeros where absent), one spare
int32_t se3d_oracle_pp_tracks(const se3d_oracle_pp *pp, int32_t *id_obs, double *state, int32_t capacity) {
  if (!pp) return -1;
  for (size_t i = 0; i < pp->tracks.size() && (int32_t)i < capacity; ++i) {
    const TrackingHypothesis &tr = pp->tracks[i];
    id_obs[2 * i] = tr.id; id_obs[2 * i + 1] = tr.num_obs;
    double *st = state + 70 * i;
    for (int q = 0; q < 70; ++q) st[q] = 0.0;
    st[0] = tr.t_prev; st[1] = tr.height_prev; st[2] = tr.root_prev.x; st[3] = tr.root_prev.y; st[4] = tr.root_prev.z;
    unsigned mask = 0u;
    for (const auto &kv : tr.prevEstimate) { mask |= 1u << kv.first; st[6 + 3 * kv.first] = kv.second.x; st[7 + 3 * kv.first] = kv.second.y; st[8 + 3 * kv.first] = kv.second.z; }
    st[5] = (double)mask;
  }
  return (int32_t)pp->tracks.size();
}
int32_t se3d_oracle_pp_last_merges(const se3d_oracle_pp *pp, int32_t *pairs, int32_t capacity_pairs) {
  if (!pp) return -1;
  for (size_t i = 0; i + 1 < pp->last_merges.size() && (int32_t)(i / 2) < capacity_pairs; i += 2) { pairs[i] = pp->last_merges[i]; pairs[i + 1] = pp->last_merges[i + 1]; }
  return (int32_t)(pp->last_merges.size() / 2);
}

// fusedSkeletonCallback (REPROJ:139-235) + draw_sigma_points (REPROJ:62-75)
int se3d_oracle_reproject(int32_t n_cam, const double *T, const double *Pm, const int32_t *wh, int32_t pose_method, int32_t n_persons,
                          const se3d_person3d *persons, se3d_reproj_out *out) {
  if (!T || !Pm || !wh || !out || (n_persons > 0 && !persons)) return SE3D_ERR_ARG;
  const int *kp2fusion = (pose_method == SE3D_POSE_H36M) ? kp2fusion_h36m : kp2fusion_simple;
  const int DIM = 3, N_SAMPLES = 7;
  std::vector<std::vector<se3d_person2d>> lists(n_cam);
  std::vector<std::vector<int>> src(n_cam);
  for (int person_idx = 0; person_idx < n_persons; ++person_idx) {
    std::vector<se3d_person2d> person_in_cam(n_cam);
    std::vector<int> num_valid(n_cam, 0);
    std::vector<double> min_x(n_cam), min_y(n_cam), max_x(n_cam, 0.0), max_y(n_cam, 0.0);
    for (int i = 0; i < n_cam; ++i) {
      std::memset(&person_in_cam[i], 0, sizeof(se3d_person2d));
      person_in_cam[i].score = 1.0f;
      min_x[i] = wh[2 * i];
      min_y[i] = wh[2 * i + 1];
    }
    for (int kp_idx = 0; kp_idx < NKP; ++kp_idx) {
      const se3d_keypoint3d &kp = persons[person_idx].keypoints[kp2fusion[kp_idx]];
      if (kp.score <= 0.0f) continue;
      const double c[9] = {kp.cov[0], kp.cov[1], kp.cov[2], kp.cov[1], kp.cov[3], kp.cov[4], kp.cov[2], kp.cov[4], kp.cov[5]};
      // cov.llt().matrixL()
      double L[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      L[0] = std::sqrt(c[0]);
      L[3] = c[3] / L[0];
      L[6] = c[6] / L[0];
      L[4] = std::sqrt(c[4] - L[3] * L[3]);
      L[7] = (c[7] - L[6] * L[3]) / L[4];
      L[8] = std::sqrt(c[8] - L[6] * L[6] - L[7] * L[7]);
      const double kappa = 0.5;
      const double b = std::sqrt(DIM + kappa);
      const double w = 1.0 / (2.0 * (DIM + kappa));  // all weights equal: 2*kappa == 1
      double samples[7][3];
      for (int a = 0; a < 3; ++a) samples[0][a] = kp.joint[a];
      for (int k = 0; k < 3; ++k)
        for (int a = 0; a < 3; ++a) {
          samples[1 + k][a] = L[a * 3 + k] * (-b) + kp.joint[a];
          samples[4 + k][a] = L[a * 3 + k] * b + kp.joint[a];
        }
      for (int i = 0; i < n_cam; ++i) {
        const double *M = T + 16 * (size_t)i;
        const double fx = Pm[4 * i], fy = Pm[4 * i + 1], cx = Pm[4 * i + 2], cy = Pm[4 * i + 3];
        double px[7], py[7];
        for (int sidx = 0; sidx < N_SAMPLES; ++sidx) {
          const double *X = samples[sidx];
          const double xc = (M[0] * X[0] + M[1] * X[1] + M[2] * X[2]) + M[3];
          const double yc = (M[4] * X[0] + M[5] * X[1] + M[6] * X[2]) + M[7];
          const double zc = (M[8] * X[0] + M[9] * X[1] + M[10] * X[2]) + M[11];
          px[sidx] = (fx * xc) / zc + cx;  // PinholeCameraModel::project3dToPixel with Tx = Ty = 0
          py[sidx] = (fy * yc) / zc + cy;
        }
        double mx = 0, my = 0;
        for (int sidx = 0; sidx < N_SAMPLES; ++sidx) { mx += px[sidx] * w; my += py[sidx] * w; }
        double cxx = 0, cxy = 0, cyy = 0;
        for (int sidx = 0; sidx < N_SAMPLES; ++sidx) {
          const double dx = px[sidx] - mx, dy = py[sidx] - my;
          cxx += dx * w * dx; cxy += dx * w * dy; cyy += dy * w * dy;
        }
        if (mx < 0 || mx > wh[2 * i] || my < 0 || my > wh[2 * i + 1]) continue;  // outside of the image plane
        if (!(mx == mx) || !(my == my)) continue;  // non-SPD covariance (NaN): skipped here; the reference would publish NaN
        ++num_valid[i];
        se3d_keypoint2d &o = person_in_cam[i].keypoints[kp_idx];
        o.x = static_cast<float>(mx); o.y = static_cast<float>(my); o.score = kp.score;
        o.cov[0] = static_cast<float>(cxx); o.cov[1] = static_cast<float>(cxy); o.cov[2] = static_cast<float>(cyy);
        if (mx < min_x[i]) min_x[i] = mx;
        if (my < min_y[i]) min_y[i] = my;
        if (mx > max_x[i]) max_x[i] = mx;
        if (my > max_y[i]) max_y[i] = my;
      }
    }
    for (int i = 0; i < n_cam; ++i)
      if (num_valid[i] > 0) {
        person_in_cam[i].bbox[0] = (float)min_x[i]; person_in_cam[i].bbox[1] = (float)min_y[i];
        person_in_cam[i].bbox[2] = (float)max_x[i]; person_in_cam[i].bbox[3] = (float)max_y[i];
        lists[i].push_back(person_in_cam[i]);
        src[i].push_back(person_idx);
      }
  }
  size_t tot = 0;
  out->det_offset[0] = 0;
  for (int i = 0; i < n_cam; ++i) { tot += lists[i].size(); out->det_offset[i + 1] = (int32_t)tot; }
  if ((int)tot > out->capacity) return SE3D_ERR_CAPACITY;
  size_t k = 0;
  for (int i = 0; i < n_cam; ++i)
    for (size_t j = 0; j < lists[i].size(); ++j, ++k) { out->persons[k] = lists[i][j]; if (out->person_index) out->person_index[k] = src[i][j]; }
  return SE3D_OK;
}

int se3d_oracle_lm(int32_t K, const double *init, int32_t n_unary, const int32_t *u_key, const double *u_meas, const double *u_cov,
                   int32_t n_range, const int32_t *r_k1, const int32_t *r_k2, const double *r_len, const double *r_sigma, double *result,
                   double *marg_cov, int32_t *iterations, double *final_error) {
  Graph g;
  Values v;
  for (int k = 0; k < K; ++k) v[k] = V3d{init[3 * k], init[3 * k + 1], init[3 * k + 2]};
  for (int i = 0; i < n_unary; ++i) {
    UnaryF u;
    u.key = u_key[i];
    for (int a = 0; a < 3; ++a) u.m[a] = u_meas[3 * i + a];
    if (!make_whitener(u_cov + 9 * (size_t)i, u.R)) return SE3D_ERR_ARG;
    g.unary.push_back(u);
  }
  for (int i = 0; i < n_range; ++i) g.range.push_back(RangeF{r_k1[i], r_k2[i], r_len[i], r_sigma[i]});
  int it = 0;
  double err = 0;
  Values res = lm_optimize(g, v, &it, &err);
  for (int k = 0; k < K; ++k) { result[3 * k] = res[k].x; result[3 * k + 1] = res[k].y; result[3 * k + 2] = res[k].z; }
  if (iterations) *iterations = it;
  if (final_error) *final_error = err;
  if (marg_cov) {
    std::map<int, std::array<double, 9>> m;
    if (!marginal_covariances(g, res, m)) return SE3D_ERR_STATE;
    for (int k = 0; k < K; ++k)
      for (int i = 0; i < 9; ++i) marg_cov[9 * (size_t)k + i] = m[k][i];
  }
  return SE3D_OK;
}

}  // extern "C"
