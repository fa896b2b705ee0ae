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
  short crow[SE3D_MUNKRES_MAX];  // step 5: compacted lists of covered rows / covered columns / uncovered columns
  short ccol[SE3D_MUNKRES_MAX];
  short ucol[SE3D_MUNKRES_MAX];
  int ncrow, nccol, nucol;
  int state;  // 0: run step 5 and continue, 1: finished, 2: aborted (iteration guard)
};

#define SE3D_DBL_EPSILON 2.2204460492503131e-16

#ifdef SE3D_LM_STAMPS  // diagnostic build only
__device__ unsigned long long g_mk_stamps[16];
#define MK_ADD(idx, v) if (threadIdx.x == 0) atomicAdd(&g_mk_stamps[idx], (unsigned long long)(v));
#define MK_CLK() __builtin_readcyclecounter()
#else
#define MK_ADD(idx, v)
#define MK_CLK() 0ull
#endif

__device__ __forceinline__ int se3d_first_bit4(unsigned long long a0, unsigned long long a1, unsigned long long a2, unsigned long long a3) {
  const int f0 = __ffsll((long long)a0) - 1, f1 = 63 + __ffsll((long long)a1), f2 = 127 + __ffsll((long long)a2), f3 = 191 + __ffsll((long long)a3);
  return a0 ? f0 : (a1 ? f1 : (a2 ? f2 : (a3 ? f3 : -1)));  // selects, no branches
}

// 64-row chunk q of the column-major matrix: column c = q / nw, row word w = q % nw (nw = ceil(nR / 64) <= 4).
// Branch-free on purpose: scalar branches per chunk would serialise the loads of the step-5 loops.
__device__ __forceinline__ void se3d_mk_chunk(int q, int nw, unsigned inv, int &c, int &w) {
  c = (int)(((unsigned)q * inv) >> 16);  // inv = ceil(65536 / nw): exact for q < 32768
  w = q - c * nw;
}
__device__ __forceinline__ unsigned long long se3d_mk_word(const unsigned long long m[4], int w) {
  return (m[0] & (0ull - (unsigned long long)(w == 0))) | (m[1] & (0ull - (unsigned long long)(w == 1))) |
         (m[2] & (0ull - (unsigned long long)(w == 2))) | (m[3] & (0ull - (unsigned long long)(w == 3)));
}

// returns 0 on success, 2 if the iteration guard fired.  assignment[r] = column or -1, r < nR.
// Force-inlined so that each call site is specialised for the address space of `dist` (LDS when the matrix fits, else HBM).
__device__ __forceinline__ int se3d_munkres_block(double *dist, int ld, int nR, int nC, MunkresShared &S, int *assignment) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (nR <= 0 || nC <= 0) return 0;
  const int minDim = nR <= nC ? nR : nC;
  const int nw = (nR + 63) >> 6;      // row words in use

  unsigned long long mk_t0 = MK_CLK();
  (void)mk_t0;
  MK_ADD(0, 1)
  for (int i = tid; i < SE3D_MUNKRES_MAX; i += 256) { S.starCol[i] = -1; S.starRow[i] = -1; S.primeCol[i] = -1; }
  for (int i = tid; i < SE3D_MUNKRES_MAX * 4; i += 256) (&S.zm[0][0])[i] = 0ull;
  __syncthreads();
  // ---- preliminary reduction (HUN:94-119 / 132-154) fused with the zero-mask build ----
  if (nR <= nC) {
    // thread r owns row r (nR <= 256): lanes of wave w hold rows 64w..64w+63, so a ballot over a column is a mask word
    const int r = tid;
    const bool act = r < nR;
    double mn = 1.7976931348623157e308;
    if (act) {
      int c = 0;
      for (; c + 4 <= nC; c += 4) {
        const double v0 = dist[r + (size_t)ld * c], v1 = dist[r + (size_t)ld * (c + 1)], v2 = dist[r + (size_t)ld * (c + 2)], v3 = dist[r + (size_t)ld * (c + 3)];
        if (v0 < mn) mn = v0;
        if (v1 < mn) mn = v1;
        if (v2 < mn) mn = v2;
        if (v3 < mn) mn = v3;
      }
      for (; c < nC; ++c) { const double v = dist[r + (size_t)ld * c]; if (v < mn) mn = v; }
    }
    if (wave < nw) {
      for (int c = 0; c < nC; ++c) {
        double v = 0.0;
        if (act) { v = dist[r + (size_t)ld * c] - mn; dist[r + (size_t)ld * c] = v; }
        const unsigned long long m = __ballot(act && fabs(v) < SE3D_DBL_EPSILON);
        if (lane == 0) S.zm[c][wave] = m;
      }
    }
  } else {
    // wave per column, two columns in flight
    for (int c0 = wave; c0 < nC; c0 += 8) {
      const int c1 = c0 + 4;
      const bool has1 = c1 < nC;
      double v0[4], v1[4];
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int r = w * 64 + lane;
        v0[w] = (w < nw && r < nR) ? dist[r + (size_t)ld * c0] : 1.7976931348623157e308;
        v1[w] = (has1 && w < nw && r < nR) ? dist[r + (size_t)ld * c1] : 1.7976931348623157e308;
      }
      double m0 = v0[0], m1 = v1[0];
#pragma unroll
      for (int w = 1; w < 4; ++w) { if (v0[w] < m0) m0 = v0[w]; if (v1[w] < m1) m1 = v1[w]; }
      for (int o = 32; o > 0; o >>= 1) {
        const double t0 = __shfl_xor(m0, o), t1 = __shfl_xor(m1, o);
        if (t0 < m0) m0 = t0;
        if (t1 < m1) m1 = t1;
      }
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        if (w < nw) {
          const int r = w * 64 + lane;
          const bool in = r < nR;
          const double a0 = v0[w] - m0;
          if (in) dist[r + (size_t)ld * c0] = a0;
          const unsigned long long b0 = __ballot(in && fabs(a0) < SE3D_DBL_EPSILON);
          if (lane == 0) S.zm[c0][w] = b0;
          if (has1) {
            const double a1 = v1[w] - m1;
            if (in) dist[r + (size_t)ld * c1] = a1;
            const unsigned long long b1 = __ballot(in && fabs(a1) < SE3D_DBL_EPSILON);
            if (lane == 0) S.zm[c1][w] = b1;
          }
        }
      }
    }
  }
  __syncthreads();

  MK_ADD(1, MK_CLK() - mk_t0)  // reduction + zero masks
  mk_t0 = MK_CLK();
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
#pragma unroll
          for (int s = 0; s < 4; ++s) {
            for (int cc = 0; cc < 64 && s * 64 + cc < nC; ++cc) {
              const int c = s * 64 + cc;
              const int rloc = se3d_first_bit4(zr[s][0] & ~covR[0], zr[s][1] & ~covR[1], zr[s][2] & ~covR[2], zr[s][3] & ~covR[3]);
              const int r = __shfl(rloc, cc);  // lane cc owns column c
              if (r >= 0) {
                if (lane == 0) { S.starCol[r] = (short)c; S.starRow[c] = (short)r; }
                covC[s] |= 1ull << cc;
                const unsigned long long rb = 1ull << (r & 63);
                const int rw = r >> 6;
                covR[0] |= (rw == 0) ? rb : 0ull; covR[1] |= (rw == 1) ? rb : 0ull; covR[2] |= (rw == 2) ? rb : 0ull; covR[3] |= (rw == 3) ? rb : 0ull;
              }
            }
          }
          covR[0] = covR[1] = covR[2] = covR[3] = 0ull;
        }
      }
      MK_ADD(7, MK_CLK() - mk_t0)  // zr reload + initial starring (first phase only)
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
        MK_ADD(5, 1)  // step-3 events
        // prime the zero (rf, cf)
        if (lane == 0) S.primeCol[rf] = (short)cf;
        int sc = S.starCol[rf];
        if (sc >= 0) {
          {  // 4-way selects instead of covR[rf >> 6] / covC[sc >> 6]: dynamic indexing would put the masks in scratch
            const unsigned long long rb = 1ull << (rf & 63), cb = ~(1ull << (sc & 63));
            const int rw = rf >> 6, cw = sc >> 6;
            covR[0] |= (rw == 0) ? rb : 0ull; covR[1] |= (rw == 1) ? rb : 0ull; covR[2] |= (rw == 2) ? rb : 0ull; covR[3] |= (rw == 3) ? rb : 0ull;
            covC[0] &= (cw == 0) ? cb : ~0ull; covC[1] &= (cw == 1) ? cb : ~0ull; covC[2] &= (cw == 2) ? cb : ~0ull; covC[3] &= (cw == 3) ? cb : ~0ull;
          }
          zerosFound = true;
          cur = cf;
        } else {
          MK_ADD(6, 1)  // augmentations
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
    MK_ADD(2, MK_CLK() - mk_t0)  // serial phase (wave 0)
    mk_t0 = MK_CLK();
    if (st != 0) break;
    MK_ADD(3, 1)  // step-5 passes
    // ---- step 5 (HUN:366-397) on all waves.  Only elements that change are touched (bit-identical to the full sweep):
    //   uncovered columns: every row  (v + h[row covered]) - h      -> zero-mask word rebuilt by ballot
    //   covered columns  : covered rows only  v + h                 -> single zero-mask bits updated
    const unsigned long long cR[4] = {S.covR[0], S.covR[1], S.covR[2], S.covR[3]};
    const unsigned long long cC[4] = {S.covC[0], S.covC[1], S.covC[2], S.covC[3]};
    if (wave < 3) {  // compacted lists: wave 0 covered rows, wave 1 covered columns, wave 2 uncovered columns
      int base = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const int idx = w * 64 + lane;
        const unsigned long long m = (wave == 0) ? cR[w] : (wave == 1) ? cC[w] : ~cC[w];
        const int lim = (wave == 0) ? nR : nC;
        const unsigned long long mv = __ballot(((m >> lane) & 1ull) && idx < lim);
        if ((mv >> lane) & 1ull) {
          const int pos = base + __popcll(mv & ((1ull << lane) - 1ull));
          if (wave == 0) S.crow[pos] = (short)idx;
          else if (wave == 1) S.ccol[pos] = (short)idx;
          else S.ucol[pos] = (short)idx;
        }
        base += __popcll(mv);
      }
      if (lane == 0) { if (wave == 0) S.ncrow = base; else if (wave == 1) S.nccol = base; else S.nucol = base; }
    }
    __syncthreads();
    const int ncr = S.ncrow, ncc = S.nccol, nuc = S.nucol;
    const int nuch = nuc * nw;  // 64-row chunks of the uncovered columns
    const unsigned inv = (65536u + (unsigned)nw - 1u) / (unsigned)nw;
    double h = 1.7976931348623157e308;  // DBL_MAX
    for (int q = wave; q < nuch; q += 4) {
      int ci, w;
      se3d_mk_chunk(q, nw, inv, ci, w);
      const int c = S.ucol[ci], r = w * 64 + lane;
      const bool ok = (r < nR) && !((se3d_mk_word(cR, w) >> lane) & 1ull);
      const double t = ok ? dist[(r < nR ? r : nR - 1) + ld * c] : 1.7976931348623157e308;
      h = t < h ? t : h;
    }
    for (int o = 32; o > 0; o >>= 1) { double t = __shfl_xor(h, o); if (t < h) h = t; }
    if (lane == 0) S.red[wave] = h;
    __syncthreads();
    h = S.red[0];
    for (int w = 1; w < 4; ++w) if (S.red[w] < h) h = S.red[w];
    // uncovered columns: "add h to each covered row", then "subtract h from each uncovered column"
    for (int q = wave; q < nuch; q += 4) {
      int ci, w;
      se3d_mk_chunk(q, nw, inv, ci, w);
      const int c = S.ucol[ci], r = w * 64 + lane;
      const bool in = r < nR;
      const double x = (dist[(in ? r : nR - 1) + ld * c] + (((se3d_mk_word(cR, w) >> lane) & 1ull) ? h : 0.0)) - h;
      if (in) dist[r + ld * c] = x;
      const unsigned long long m = __ballot(in && (fabs(x) < SE3D_DBL_EPSILON));
      if (lane == 0) S.zm[c][w] = m;
    }
    // covered columns change only at covered rows: "add h to each covered row"; items = (covered row, covered column)
    {
      int lg = 0;
      while ((1 << lg) < ncr) ++lg;            // rows per group: R = 2^lg >= ncr
      const int R = 1 << lg, i = tid & (R - 1), j0 = tid >> lg, jstep = 256 >> lg;
      if (i < ncr && jstep > 0) {
        const int r = S.crow[i];
        const unsigned long long bit = 1ull << (r & 63);
        for (int j = j0; j < ncc; j += jstep) {
          const int c = S.ccol[j];
          const double x = dist[r + ld * c] + h;
          dist[r + ld * c] = x;
          if (fabs(x) < SE3D_DBL_EPSILON) atomicOr(&S.zm[c][r >> 6], bit);
          else atomicAnd(&S.zm[c][r >> 6], ~bit);
        }
      }
    }
    __syncthreads();
    MK_ADD(4, MK_CLK() - mk_t0)  // step-5 cycles
    mk_t0 = MK_CLK();
  }
  const int fin = S.state;
  for (int r = tid; r < nR; r += 256) assignment[r] = S.starCol[r];  // buildassignmentvector (HUN:190-205)
  __syncthreads();
  return fin == 2 ? 2 : 0;
}

// =============================================================================================================
// Single-wavefront variant for nR, nC <= 64 (small association problems: 16 cameras x <= 8 detections, track association
// of a handful of persons).  Same algorithm, same scan orders, same fp64 arithmetic per element as se3d_munkres_block —
// but lane c owns column c (zero-row mask in a register), lane r owns row r (starCol / primeCol in registers, starRow on
// the column lane), cover sets are uniform 64-bit masks and there is no workgroup barrier anywhere, so a frame / sequence
// needs one wavefront instead of four and its latency-bound control flow leaves the other wave slots of the CU free.
// dist: column-major (row + ld * col), modified in place; lane r touches row r only.  On return lane r < nR holds its column
// (or -1) in assign_lane.  Returns 0, or 2 if the iteration guard fired.
// =============================================================================================================
__device__ __forceinline__ double se3d_wave_min_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { const double t = __shfl_xor(v, o); v = t < v ? t : v; }
  return v;
}

__device__ __forceinline__ int se3d_munkres_wave(double *dist, int ld, int nR, int nC, int lane, int &assign_lane) {
  assign_lane = -1;
  if (nR <= 0 || nC <= 0) return 0;
  const int minDim = nR <= nC ? nR : nC;
  const unsigned long long colsMask = nC >= 64 ? ~0ull : ((1ull << nC) - 1ull);
  const bool inR = lane < nR;
  const double DMAX = 1.7976931348623157e308;
  unsigned long long zr = 0ull;  // lane c: rows of column c that hold a zero
  int starCol = -1, primeCol = -1;  // lane r
  int starRow = -1;                 // lane c
  // ---- preliminary reduction (HUN:94-119 / 132-154) + zero masks ----
  if (nR <= nC) {
    double mn = DMAX;
    if (inR)
      for (int c = 0; c < nC; ++c) { const double v = dist[lane + ld * c]; mn = v < mn ? v : mn; }
    for (int c = 0; c < nC; ++c) {
      double v = 1.0;
      if (inR) { v = dist[lane + ld * c] - mn; dist[lane + ld * c] = v; }
      const unsigned long long m = __ballot(inR && fabs(v) < SE3D_DBL_EPSILON);
      if (lane == c) zr = m;
    }
  } else {
    for (int c = 0; c < nC; ++c) {
      const double v = inR ? dist[lane + ld * c] : DMAX;
      const double m0 = se3d_wave_min_f64(v);
      const double a = v - m0;
      if (inR) dist[lane + ld * c] = a;
      const unsigned long long m = __ballot(inR && fabs(a) < SE3D_DBL_EPSILON);
      if (lane == c) zr = m;
    }
  }
  // ---- steps 1 / 2a: initial starring (HUN:121-130 / 156-168) ----
  unsigned long long covR = 0ull, covC = 0ull;
  if (nR <= nC) {
    for (int r = 0; r < nR; ++r) {
      const bool cand = ((zr >> r) & 1ull) && !((covC >> lane) & 1ull);
      const unsigned long long b = __ballot(cand);
      if (b) {
        const int c = __ffsll((long long)b) - 1;
        if (lane == r) starCol = c;
        if (lane == c) starRow = r;
        covC |= 1ull << c;
      }
    }
  } else {
    for (int c = 0; c < nC; ++c) {
      const int rloc = __ffsll((long long)(zr & ~covR)) - 1;
      const int r = __builtin_amdgcn_readlane(rloc, c);
      if (r >= 0) {
        if (lane == r) starCol = c;
        if (lane == c) starRow = r;
        covC |= 1ull << c;
        covR |= 1ull << r;
      }
    }
    covR = 0ull;
  }
  int guard = 0, st = -1;
  for (;;) {
    // ---- step 2b / 3 / 4 (HUN:245-363) ----
    st = -1;
    bool need2b = true, zerosFound = false;
    int cur = -1;
    while (st < 0) {
      if (need2b) {
        if (__popcll(covC) == minDim) { st = 1; break; }
        need2b = false;
        cur = -1;
        zerosFound = false;
      }
      if (++guard > 4000000) { st = 2; break; }
      const unsigned long long m = zr & ~covR;
      const bool cand = (lane < nC) && (lane > cur) && !((covC >> lane) & 1ull) && (m != 0ull);
      const unsigned long long b = __ballot(cand);
      if (!b) {
        if (zerosFound) { zerosFound = false; cur = -1; continue; }
        st = 0;
        break;
      }
      const int cf = __ffsll((long long)b) - 1;
      const int rf = __builtin_amdgcn_readlane(__ffsll((long long)m) - 1, cf);
      if (lane == rf) primeCol = cf;
      const int sc = __builtin_amdgcn_readlane(starCol, rf);
      if (sc >= 0) {
        covR |= 1ull << rf;
        covC &= ~(1ull << sc);
        zerosFound = true;
        cur = cf;
      } else {  // step 4: augment along the alternating path
        int row = rf, col = cf;
        for (int it = 0; it < 2 * 64 + 2; ++it) {
          const int sr = __builtin_amdgcn_readlane(starRow, col);
          if (lane == row) starCol = col;
          if (lane == col) starRow = row;
          if (sr < 0) break;
          const int pc = __builtin_amdgcn_readlane(primeCol, sr);
          row = sr;
          col = pc;
        }
        primeCol = -1;
        covR = 0ull;
        covC = __ballot(lane < nC && starRow >= 0);  // step 2a
        need2b = true;
      }
    }
    if (st != 0) break;
    // ---- step 5 (HUN:366-397): only the elements that change ----
    const bool rowCov = inR && ((covR >> lane) & 1ull);
    const bool rowUnc = inR && !rowCov;
    double h = DMAX;
    for (unsigned long long cm = ~covC & colsMask; cm; cm &= cm - 1ull) {
      const int c = __ffsll((long long)cm) - 1;
      const double t = rowUnc ? dist[lane + ld * c] : DMAX;
      h = t < h ? t : h;
    }
    h = se3d_wave_min_f64(h);
    const double hadd = rowCov ? h : 0.0;
    for (unsigned long long cm = ~covC & colsMask; cm; cm &= cm - 1ull) {  // uncovered columns: every row
      const int c = __ffsll((long long)cm) - 1;
      double x = 1.0;
      if (inR) { x = (dist[lane + ld * c] + hadd) - h; dist[lane + ld * c] = x; }
      const unsigned long long mz = __ballot(inR && fabs(x) < SE3D_DBL_EPSILON);
      if (lane == c) zr = mz;
    }
    const unsigned long long mcov = __ballot(rowCov);
    for (unsigned long long cm = covC & colsMask; cm; cm &= cm - 1ull) {   // covered columns: covered rows only
      const int c = __ffsll((long long)cm) - 1;
      double x = 1.0;
      if (rowCov) { x = dist[lane + ld * c] + h; dist[lane + ld * c] = x; }
      const unsigned long long mz = __ballot(rowCov && fabs(x) < SE3D_DBL_EPSILON);
      if (lane == c) zr = (zr & ~mcov) | mz;
    }
  }
  assign_lane = inR ? starCol : -1;
  return st == 2 ? 2 : 0;
}
