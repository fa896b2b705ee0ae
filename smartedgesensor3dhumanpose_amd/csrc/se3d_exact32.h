// se3d_exact32.h — the oracle's restatement of the reference's float32 DLT solve, operation for operation (round 4).
//
// skeleton_3d takes the null vector of every DLT system — the published point (S3D:746), each leave-one-view-out candidate (S3D:793-838)
// and each of the 4n + 1 sigma samples of the unscented transform (S3D:518) — from Eigen 3.3's JacobiSVD in FLOAT (S3D:456:
// `A.jacobiSvd(ComputeThinV).matrixV().col(3)`).  The fast triangulation kernels solve the same float32 rows by fp64 inverse iteration, which
// agrees with that to c * 2^-24 * kappa metres (kappa = (1 + |X|^2) s1 / (s3 - s4); c <= 4.9 over 3e5 joints): far below the 1 mm contract
// on a well-conditioned joint, metres apart on an ill-conditioned one, and a different branch of the outlier rejection when a reprojection
// error sits at a threshold.  Joints the fast kernels flag (se3d_kappa_bound >= SE3D_EXACT_KAPPA, reprojection error above the guard band of
// g_reproj_error_max_acceptable, a solve that fell back) are redone HERE with the arithmetic of the reference: JacobiSVD::compute of a
// rows x 4 float matrix = scale by the largest |entry|, ColPivHouseholderQR preconditioner when rows != 4 (column norms, pivoting, Householder
// reflections with the LAPACK norm downdate), two-sided Jacobi sweeps on the 4 x 4 factor with real_2x2_jacobi_svd / makeJacobi, descending
// sort of the singular values, last column of V — every sum sequential, no FMA contraction (the file is compiled with -ffp-contract=off), IEEE
// division and square root.  All of it is IEEE-754 binary32 arithmetic in a fixed order, so the result equals, bit for bit, what the same
// sequence gives on any IEEE machine; tests/test_gpu_exact32.py checks that against the CPU restatement the parity tests use.
//
// One solve runs on ONE lane (the sums are sequential by definition); a wavefront runs up to 64 independent solves — the full system and its
// leave-one-out subsets, then the sigma samples.  The lane's rows x 4 matrix lives in a lane-interleaved workspace (element e of lane l at
// base[e * 64 + l]: coalesced), the 4 x 4 factor and V in registers.
#pragma once
#include <float.h>
#include <hip/hip_runtime.h>

// lane-private rows x 4 float matrix, column-major with leading dimension ld; element (i, c) at p[(c * ld + i) * stride]
template <typename PT>
struct X32MatT {
  PT p;
  int stride, ld;
  __device__ __forceinline__ float get(int i, int c) const { return p[(size_t)(c * ld + i) * stride]; }
  __device__ __forceinline__ void set(int i, int c, float v) const { p[(size_t)(c * ld + i) * stride] = v; }
};
typedef X32MatT<float *> X32Mat;

// JacobiRotation::makeJacobi(x, y, z) (Eigen/src/Jacobi/Jacobi.h)
__device__ __forceinline__ void x32_make_jacobi(float x, float y, float z, float &c, float &s) {
  const float deno = 2.0f * fabsf(y);
  if (deno < FLT_MIN) { c = 1.f; s = 0.f; return; }
  const float tau = (x - z) / deno;
  const float w = sqrtf(tau * tau + 1.0f);
  const float t = (tau > 0.f) ? 1.0f / (tau + w) : 1.0f / (tau - w);
  const float sign_t = t > 0.f ? 1.f : -1.f;
  const float n = 1.0f / sqrtf(t * t + 1.0f);
  s = -sign_t * (y / fabsf(y)) * fabsf(t) * n;
  c = n;
}

// JacobiSVD::compute steps 2 - 4 on the 4 x 4 work matrix W with the accumulated right factor V: v_out = V.col(3) after the descending sort.
// (p, q) loops fully unrolled so that W and V stay in registers.
__device__ __forceinline__ void x32_jacobi4(float (&W)[4][4], float (&V)[4][4], float v_out[4]) {
  const float precision = 2.0f * FLT_EPSILON;
  const float considerAsZero = FLT_MIN;
  float maxDiag = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) maxDiag = fmaxf(maxDiag, fabsf(W[i][i]));
  bool finished = false;
  int guard = 0;
  while (!finished && guard++ < 200) {
    finished = true;
#pragma unroll
    for (int p = 1; p < 4; ++p) {
#pragma unroll
      for (int q = 0; q < p; ++q) {
        const float threshold = fmaxf(considerAsZero, precision * maxDiag);
        if (fabsf(W[p][q]) > threshold || fabsf(W[q][p]) > threshold) {
          finished = false;
          // real_2x2_jacobi_svd (Eigen/src/SVD/JacobiSVD.h)
          const float m00 = W[p][p], m01 = W[p][q], m10 = W[q][p], m11 = W[q][q];
          float r1c, r1s;
          const float t = m00 + m11, d = m10 - m01;
          if (fabsf(d) < FLT_MIN) {
            r1s = 0.f; r1c = 1.f;
          } else {
            const float u = t / d;
            const float tm = sqrtf(1.0f + u * u);
            r1s = 1.0f / tm;
            r1c = u / tm;
          }
          const float n00 = r1c * m00 + r1s * m10, n01 = r1c * m01 + r1s * m11;
          const float n11 = -r1s * m01 + r1c * m11;
          float jrc, jrs;
          x32_make_jacobi(n00, n01, n11, jrc, jrs);
          const float jtc = jrc, jts = -jrs;                                      // j_right.transpose()
          const float jlc = r1c * jtc - r1s * jts, jls = r1c * jts + r1s * jtc;   // j_left = rot1 * j_right.transpose()
#pragma unroll
          for (int j = 0; j < 4; ++j) {   // applyOnTheLeft(p, q, j_left)
            const float x = W[p][j], y = W[q][j];
            W[p][j] = jlc * x + jls * y;
            W[q][j] = -jls * x + jlc * y;
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {   // applyOnTheRight(p, q, j_right)
            const float x = W[i][p], y = W[i][q];
            W[i][p] = jrc * x - jrs * y;
            W[i][q] = jrs * x + jrc * y;
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float x = V[i][p], y = V[i][q];
            V[i][p] = jrc * x - jrs * y;
            V[i][q] = jrs * x + jrc * y;
          }
          maxDiag = fmaxf(maxDiag, fmaxf(fabsf(W[p][p]), fabsf(W[q][q])));
        }
      }
    }
  }
  // singular values |W_ii|, descending selection sort with column swaps: only the column that ends in position 3 matters
  float sv[4];
  int idx[4] = {0, 1, 2, 3};
#pragma unroll
  for (int i = 0; i < 4; ++i) sv[i] = fabsf(W[i][i]);
  bool stop = false;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (stop) break;
    int pos = i;
    float best = sv[i];
#pragma unroll
    for (int j = i + 1; j < 4; ++j)
      if (sv[j] > best) { best = sv[j]; pos = j; }
    if (best == 0.f) { stop = true; break; }
#pragma unroll
    for (int j = i + 1; j < 4; ++j)
      if (pos == j) {
        const float ts = sv[i]; sv[i] = sv[j]; sv[j] = ts;
        const int ti = idx[i]; idx[i] = idx[j]; idx[j] = ti;
      }
  }
  const int m = idx[3];
#pragma unroll
  for (int i = 0; i < 4; ++i) v_out[i] = (m == 0) ? V[i][0] : (m == 1) ? V[i][1] : (m == 2) ? V[i][2] : V[i][3];
}

// Row loops of the QR step: the lane's column lives in memory (global workspace), every sum is sequential by definition, but the LOADS of
// X32_BS consecutive rows are independent — they are issued together and the dependent float chain runs on registers (one memory round trip
// per X32_BS rows instead of per row: the second pass was bound by that latency, 176 ms per 64 frames at 256 cameras; profiles/r4/README.md).
// A masked tail element contributes nothing (the accumulation is skipped, not fed a zero: a zero could flip the sign of a zero sum).
#define X32_BS 8
template <typename PT>
__device__ __forceinline__ float x32_sumsq(const X32MatT<PT> &A, int c, int i0, int rows) {   // sum_{i = i0}^{rows-1} A(i, c)^2, ascending
  float ss = 0.f;
  for (int i = i0; i < rows; i += X32_BS) {
    float a[X32_BS];
#pragma unroll
    for (int u = 0; u < X32_BS; ++u) a[u] = (i + u < rows) ? A.get(i + u, c) : 0.f;
#pragma unroll
    for (int u = 0; u < X32_BS; ++u) ss = (i + u < rows) ? ss + a[u] * a[u] : ss;
  }
  return ss;
}

// Null vector of the rows x 4 system in A (rows >= 4; A is overwritten): JacobiSVD<Matrix<float, -1, 4>>::compute + matrixV().col(3).
// Out of line on purpose (one copy of ~3 k instructions for the three call sites of the second pass; arguments and result in registers).
// `amax`: the largest |entry| of A (the builder of the matrix tracks it while it writes the rows: one pass over the matrix less).
template <typename PT>
__device__ __noinline__ float4 x32_null_vector(int rows, PT Ap, int Astride, int Ald, float amax) {
  const X32MatT<PT> A{Ap, Astride, Ald};
  float v_out[4];
  float scale = amax;   // JacobiSVD::compute: scale = matrix.cwiseAbs().maxCoeff(); if (scale == 0) scale = 1
  if (scale == 0.f) scale = 1.f;
  // matrix / scale, fused with the column norms of the QR step below (sums over the scaled entries in row order: the same values, the same order)
  float colss[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 4; ++c)
    for (int i = 0; i < rows; i += X32_BS) {
      float a[X32_BS];
#pragma unroll
      for (int u = 0; u < X32_BS; ++u) a[u] = (i + u < rows) ? A.get(i + u, c) : 0.f;
#pragma unroll
      for (int u = 0; u < X32_BS; ++u)
        if (i + u < rows) {
          const float q = a[u] / scale;
          A.set(i + u, c, q);
          colss[c] += q * q;
        }
    }
  float W[4][4], V[4][4];
  if (rows == 4) {   // square input: no QR step (m_workMatrix = matrix / scale, m_matrixV.setIdentity())
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) { W[i][j] = A.get(i, j); V[i][j] = (i == j) ? 1.f : 0.f; }
    x32_jacobi4(W, V, v_out);
    return make_float4(v_out[0], v_out[1], v_out[2], v_out[3]);
  }
  // ---- ColPivHouseholderQR (Eigen/src/QR/ColPivHouseholderQR.h); the column swaps are carried by perm (logical column k = stored column perm[k]) ----
  int perm[4] = {0, 1, 2, 3};
  float nu[4], nd[4];   // m_colNormsUpdated, m_colNormsDirect
#pragma unroll
  for (int k = 0; k < 4; ++k) nu[k] = nd[k] = sqrtf(colss[k]);
  const float norm_downdate_threshold = sqrtf(FLT_EPSILON);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int big = k;
    float bigv = nu[k];
#pragma unroll
    for (int j = k + 1; j < 4; ++j)
      if (nu[j] > bigv) { bigv = nu[j]; big = j; }
#pragma unroll
    for (int j = k + 1; j < 4; ++j)
      if (big == j) {
        const float t0 = nu[k]; nu[k] = nu[j]; nu[j] = t0;
        const float t1 = nd[k]; nd[k] = nd[j]; nd[j] = t1;
        const int t2 = perm[k]; perm[k] = perm[j]; perm[j] = t2;
      }
    const int pk = perm[k];
    // makeHouseholderInPlace on A.col(k).tail(rows - k)
    const float tailSq = x32_sumsq(A, pk, k + 1, rows);
    const float c0 = A.get(k, pk);
    float tau, beta, den = 1.f;   // den: the essential part (tail / den) is formed by the pass that takes the dot products below
    if (tailSq <= FLT_MIN) {
      tau = 0.f;
      beta = c0;
      for (int i = k + 1; i < rows; ++i) A.set(i, pk, 0.f);
    } else {
      beta = sqrtf(c0 * c0 + tailSq);
      if (c0 >= 0.f) beta = -beta;
      den = c0 - beta;
      tau = (beta - c0) / beta;
    }
    A.set(k, pk, beta);
    // applyHouseholderOnTheLeft to bottomRightCorner(rows - k, 3 - k)  (rows - k >= 3 here: rows >= 6)
    // (the last column's essential part is never read again — only R enters the Jacobi step — so its scaling is skipped)
    if (tau != 0.f && k + 1 < 4) {
      // tmp[j] = essential^T * column j (sequential over the rows) + A(k, j); the reflector's column is scaled (tail / den, stored) and used
      // in the same pass, loaded once per batch of rows for all trailing columns
      float tmp[4] = {0.f, 0.f, 0.f, 0.f};
      for (int i = k + 1; i < rows; i += X32_BS) {
        float vk[X32_BS];
#pragma unroll
        for (int u = 0; u < X32_BS; ++u) vk[u] = (i + u < rows) ? A.get(i + u, pk) : 0.f;
#pragma unroll
        for (int u = 0; u < X32_BS; ++u)
          if (i + u < rows) { vk[u] = vk[u] / den; A.set(i + u, pk, vk[u]); }
#pragma unroll
        for (int j = k + 1; j < 4; ++j) {
          const int pj = perm[j];
          float vj[X32_BS];
#pragma unroll
          for (int u = 0; u < X32_BS; ++u) vj[u] = (i + u < rows) ? A.get(i + u, pj) : 0.f;
#pragma unroll
          for (int u = 0; u < X32_BS; ++u) tmp[j] = (i + u < rows) ? tmp[j] + vk[u] * vj[u] : tmp[j];
        }
      }
#pragma unroll
      for (int j = k + 1; j < 4; ++j) {
        const int pj = perm[j];
        tmp[j] += A.get(k, pj);
        A.set(k, pj, A.get(k, pj) - tau * tmp[j]);
      }
      for (int i = k + 1; i < rows; i += X32_BS) {
        float vk[X32_BS];
#pragma unroll
        for (int u = 0; u < X32_BS; ++u) vk[u] = (i + u < rows) ? A.get(i + u, pk) : 0.f;
#pragma unroll
        for (int j = k + 1; j < 4; ++j) {
          const int pj = perm[j];
          float vj[X32_BS];
#pragma unroll
          for (int u = 0; u < X32_BS; ++u) vj[u] = (i + u < rows) ? A.get(i + u, pj) : 0.f;
#pragma unroll
          for (int u = 0; u < X32_BS; ++u)
            if (i + u < rows) A.set(i + u, pj, vj[u] - tau * vk[u] * tmp[j]);
        }
      }
    }
    // LAPACK-style column-norm downdate (lawn176)
#pragma unroll
    for (int j = k + 1; j < 4; ++j) {
      if (nu[j] != 0.f) {
        const int pj = perm[j];
        float temp = fabsf(A.get(k, pj)) / nu[j];
        temp = (1.0f + temp) * (1.0f - temp);
        temp = temp < 0.f ? 0.f : temp;
        const float r = nu[j] / nd[j];
        const float temp2 = temp * (r * r);
        if (temp2 <= norm_downdate_threshold) {
          nd[j] = sqrtf(x32_sumsq(A, pj, k + 1, rows));
          nu[j] = nd[j];
        } else {
          nu[j] *= sqrtf(temp);
        }
      }
    }
  }
  // work matrix = R (upper triangular 4 x 4), V = the column permutation
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      W[i][j] = (j >= i) ? A.get(i, perm[j]) : 0.f;
      V[i][j] = (perm[j] == i) ? 1.f : 0.f;
    }
  x32_jacobi4(W, V, v_out);
  return make_float4(v_out[0], v_out[1], v_out[2], v_out[3]);
}

// rows of the DLT system for one view exactly as S3D:446-449: r = coord * P.row(2) - P.row(k), then row.normalize() (division by the norm)
__device__ __forceinline__ void x32_dlt_rows(const float *__restrict__ M, float x, float y, float r0[4], float r1[4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) { r0[k] = x * M[8 + k] - M[k]; r1[k] = y * M[8 + k] - M[4 + k]; }
  const float z0 = (r0[0] * r0[0] + r0[1] * r0[1]) + (r0[2] * r0[2] + r0[3] * r0[3]);
  if (z0 > 0.f) { const float nn = sqrtf(z0);
#pragma unroll
    for (int k = 0; k < 4; ++k) r0[k] = r0[k] / nn; }
  const float z1 = (r1[0] * r1[0] + r1[1] * r1[1]) + (r1[2] * r1[2] + r1[3] * r1[3]);
  if (z1 > 0.f) { const float nn = sqrtf(z1);
#pragma unroll
    for (int k = 0; k < 4; ++k) r1[k] = r1[k] / nn; }
}
