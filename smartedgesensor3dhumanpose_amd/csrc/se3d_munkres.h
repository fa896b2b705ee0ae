// se3d_munkres.h — block-cooperative Munkres assignment for gfx950, bit-exact with the reference's
// HungarianAlgorithm::assignmentoptimal (skeleton_3d/src/Hungarian.cpp:60-397; called at S3D:630 and PP:561).
//
// The reference is a chain of mutually tail-recursive steps over boolean star/prime/cover matrices with
// column-major scans whose ORDER decides ties.  This version keeps every scan order but re-expresses the state for a
// 64-wide wavefront:
//   * zero matrix  -> per-column row bit-masks zm[c][4] (256 rows); a column scan "first uncovered zero row" is
//                     ffs(zm[c] & ~coveredRows); the column-major search of step 3 (HUN:278-304) is a ballot over
//                     lanes that each own 4 columns, so one pass costs O(events) wave instructions, not O(nR*nC);
//   * star/prime   -> starCol[row], starRow[col], primeCol[row] (a row holds at most one star and, between clears,
//                     at most one prime), step 4 (HUN:312-363) walks the alternating path through these arrays;
//   * step 5       -> (HUN:366-397) the only O(nR*nC) arithmetic (fp64 add/sub/compare: exact, order-free) runs on
//                     all 4 wavefronts of the block and rebuilds the zero masks with ballots.
// Wave 0 runs the serial control flow with cover masks in uniform registers; other waves join at barriers.
// blockDim.x must be 256.  dist is column-major (row + ld*col) and is modified in place.
#pragma once
#include "se3d_device.h"

struct MunkresShared {
  unsigned long long zm[SE3D_MUNKRES_MAX][4];
  short starCol[SE3D_MUNKRES_MAX];
  short starRow[SE3D_MUNKRES_MAX];
  short primeCol[SE3D_MUNKRES_MAX];
  unsigned long long covR[4], covC[4];
  double red[4];
  int state;  // 0: run step 5 and continue, 1: finished, 2: aborted (iteration guard)
};

#define SE3D_DBL_EPSILON 2.2204460492503131e-16

__device__ __forceinline__ int se3d_first_bit4(unsigned long long a0, unsigned long long a1, unsigned long long a2, unsigned long long a3) {
  if (a0) return __ffsll((long long)a0) - 1;
  if (a1) return 64 + __ffsll((long long)a1) - 1;
  if (a2) return 128 + __ffsll((long long)a2) - 1;
  if (a3) return 192 + __ffsll((long long)a3) - 1;
  return -1;
}

// returns 0 on success, 2 if the iteration guard fired.  assignment[r] = column or -1, r < nR.
__device__ int se3d_munkres_block(double *dist, int ld, int nR, int nC, MunkresShared &S, int *assignment) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (nR <= 0 || nC <= 0) return 0;
  const int minDim = nR <= nC ? nR : nC;

  // ---- preliminary reduction (HUN:94-119 / 132-154) ----
  if (nR <= nC) {
    for (int r = tid; r < nR; r += 256) {
      double mn = dist[r];
      for (int c = 1; c < nC; ++c) { double v = dist[r + (size_t)ld * c]; if (v < mn) mn = v; }
      for (int c = 0; c < nC; ++c) dist[r + (size_t)ld * c] -= mn;
    }
  } else {
    for (int c = wave; c < nC; c += 4) {
      double mn = 1.7976931348623157e308;
      for (int r = lane; r < nR; r += 64) { double v = dist[r + (size_t)ld * c]; if (v < mn) mn = v; }
      for (int o = 32; o > 0; o >>= 1) { double t = __shfl_xor(mn, o); if (t < mn) mn = t; }
      for (int r = lane; r < nR; r += 64) dist[r + (size_t)ld * c] -= mn;
    }
  }
  for (int i = tid; i < SE3D_MUNKRES_MAX; i += 256) { S.starCol[i] = -1; S.starRow[i] = -1; S.primeCol[i] = -1; }
  __syncthreads();
  // zero masks
  for (int c = wave; c < nC; c += 4)
    for (int w = 0; w < 4; ++w) {
      int r = w * 64 + lane;
      bool z = (r < nR) && (fabs(dist[r + (size_t)ld * c]) < SE3D_DBL_EPSILON);
      unsigned long long m = __ballot(z);
      if (lane == 0) S.zm[c][w] = m;
    }
  __syncthreads();

  unsigned long long covR[4] = {0, 0, 0, 0}, covC[4] = {0, 0, 0, 0};
  unsigned long long zr[4][4];
  bool first = true;
  int guard = 0;
  for (;;) {
    if (wave == 0) {
      // (re)load this lane's 4 columns of the zero matrix
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        int c = s * 64 + lane;
#pragma unroll
        for (int w = 0; w < 4; ++w) zr[s][w] = (c < nC) ? S.zm[c][w] : 0ull;
      }
      if (first) {
        first = false;
        // ---- steps 1 and 2a: initial starring (HUN:121-130 / 156-168) ----
        if (nR <= nC) {
          for (int r = 0; r < nR; ++r) {
            const int w = r >> 6;
            const unsigned long long bit = 1ull << (r & 63);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
              unsigned long long zw = (w == 0) ? zr[s][0] : (w == 1) ? zr[s][1] : (w == 2) ? zr[s][2] : zr[s][3];
              bool cand = (zw & bit) && !((covC[s] >> lane) & 1ull);
              unsigned long long b = __ballot(cand);
              if (b) {
                int fl = __ffsll((long long)b) - 1;
                int c = s * 64 + fl;
                if (lane == 0) { S.starCol[r] = (short)c; S.starRow[c] = (short)r; }
                covC[s] |= 1ull << fl;
                break;
              }
            }
          }
        } else {
          for (int c = 0; c < nC; ++c) {
            unsigned long long m0 = S.zm[c][0] & ~covR[0], m1 = S.zm[c][1] & ~covR[1], m2 = S.zm[c][2] & ~covR[2], m3 = S.zm[c][3] & ~covR[3];
            int r = se3d_first_bit4(m0, m1, m2, m3);
            if (r >= 0) {
              if (lane == 0) { S.starCol[r] = (short)c; S.starRow[c] = (short)r; }
              covC[c >> 6] |= 1ull << (c & 63);
              covR[r >> 6] |= 1ull << (r & 63);
            }
          }
          covR[0] = covR[1] = covR[2] = covR[3] = 0ull;
        }
      }
      // ---- step 2b / 3 / 4 loop (HUN:245-363) ----
      int st = -1;
      bool need2b = true;
      int cur = -1;
      bool zerosFound = false;
      while (st < 0) {
        if (need2b) {
          int ncov = __popcll(covC[0]) + __popcll(covC[1]) + __popcll(covC[2]) + __popcll(covC[3]);
          if (ncov == minDim) { st = 1; break; }
          need2b = false;
          cur = -1;
          zerosFound = false;
        }
        if (++guard > 4000000) { st = 2; break; }
        // find first uncovered column > cur that holds an uncovered zero
        int cf = -1, rf = -1;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          if (cf < 0) {
            int c = s * 64 + lane;
            unsigned long long m0 = zr[s][0] & ~covR[0], m1 = zr[s][1] & ~covR[1], m2 = zr[s][2] & ~covR[2], m3 = zr[s][3] & ~covR[3];
            bool cand = (c < nC) && (c > cur) && !((covC[s] >> lane) & 1ull) && ((m0 | m1 | m2 | m3) != 0ull);
            unsigned long long b = __ballot(cand);
            if (b) {
              int fl = __ffsll((long long)b) - 1;
              int rloc = se3d_first_bit4(m0, m1, m2, m3);
              rf = __shfl(rloc, fl);
              cf = s * 64 + fl;
            }
          }
        }
        if (cf < 0) {
          if (zerosFound) { zerosFound = false; cur = -1; continue; }  // next pass of while(zerosFound)
          st = 0;  // step 5
          break;
        }
        // prime the zero (rf, cf)
        if (lane == 0) S.primeCol[rf] = (short)cf;
        int sc = S.starCol[rf];
        if (sc >= 0) {
          covR[rf >> 6] |= 1ull << (rf & 63);
          covC[sc >> 6] &= ~(1ull << (sc & 63));
          zerosFound = true;
          cur = cf;
        } else {
          // step 4: augment along the alternating path
          int row = rf, col = cf;
          for (int it = 0; it < 2 * SE3D_MUNKRES_MAX + 2; ++it) {
            int sr = S.starRow[col];
            if (lane == 0) { S.starCol[row] = (short)col; S.starRow[col] = (short)row; }
            if (sr < 0) break;
            int pc = S.primeCol[sr];
            row = sr;
            col = pc;
          }
#pragma unroll
          for (int s = 0; s < 4; ++s) S.primeCol[s * 64 + lane] = -1;
          covR[0] = covR[1] = covR[2] = covR[3] = 0ull;
#pragma unroll
          for (int s = 0; s < 4; ++s) {  // step 2a
            int c = s * 64 + lane;
            covC[s] = __ballot(c < nC && S.starRow[c] >= 0);
          }
          need2b = true;
        }
      }
      if (lane == 0) {
        for (int w = 0; w < 4; ++w) { S.covR[w] = covR[w]; S.covC[w] = covC[w]; }
        S.state = st;
      }
    }
    __syncthreads();
    const int st = S.state;
    if (st != 0) break;
    // ---- step 5 (HUN:366-397) on all waves ----
    unsigned long long cR[4] = {S.covR[0], S.covR[1], S.covR[2], S.covR[3]};
    double h = 1.7976931348623157e308;  // DBL_MAX
    for (int c = wave; c < nC; c += 4) {
      if ((S.covC[c >> 6] >> (c & 63)) & 1ull) continue;
      for (int w = 0; w < 4; ++w) {
        int r = w * 64 + lane;
        if (r < nR && !((cR[w] >> lane) & 1ull)) { double v = dist[r + (size_t)ld * c]; if (v < h) h = v; }
      }
    }
    for (int o = 32; o > 0; o >>= 1) { double t = __shfl_xor(h, o); if (t < h) h = t; }
    if (lane == 0) S.red[wave] = h;
    __syncthreads();
    h = S.red[0];
    for (int w = 1; w < 4; ++w) if (S.red[w] < h) h = S.red[w];
    for (int c = wave; c < nC; c += 4) {
      const bool cc = (S.covC[c >> 6] >> (c & 63)) & 1ull;
      for (int w = 0; w < 4; ++w) {
        int r = w * 64 + lane;
        bool z = false;
        if (r < nR) {
          double v = dist[r + (size_t)ld * c];
          const bool rc = (cR[w] >> lane) & 1ull;
          if (rc) v = v + h;    // "add h to each covered row"
          if (!cc) v = v - h;   // "subtract h from each uncovered column"
          if (rc || !cc) dist[r + (size_t)ld * c] = v;
          z = fabs(v) < SE3D_DBL_EPSILON;
        }
        unsigned long long m = __ballot(z);
        if (lane == 0) S.zm[c][w] = m;
      }
    }
    __syncthreads();
  }
  const int fin = S.state;
  for (int r = tid; r < nR; r += 256) assignment[r] = S.starCol[r];  // buildassignmentvector (HUN:190-205)
  __syncthreads();
  return fin == 2 ? 2 : 0;
}
