/*
 * se3d_oracle.h — CPU ORACLE (TEST INFRASTRUCTURE ONLY).
 *
 * Plain-C++ restatement of the reference hot path, function-for-function with SURVEY.md §8a.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the checker /
 * reported baseline.  The product library (libse3d_hip.so) never links or calls it.
 *
 * Pinning status:
 *   - Hungarian (row M): PINNED bit-exactly against the reference's own Hungarian.cpp compiled from
 *     /root/reference into oracle/_ref/ (tests/golden/hungarian_*.json were produced by that binary).
 *   - Everything that goes through Eigen 3 JacobiSVD / GTSAM 4.0.3 (rows T, U, R): PARITY UNPINNED — those
 *     libraries are not in the reference tree and not in this image; the algorithms are restated from their
 *     published behaviour and checked against independent numpy/scipy solutions (tests/test_oracle_kat.py).
 */
#ifndef SE3D_ORACLE_H
#define SE3D_ORACLE_H
#include "../include/se3d.h"

#ifdef __cplusplus
extern "C" {
#endif

/* S3D:1187-1211 + 230-253: Rt (float) and fundamental matrices (double compute, float store). */
int se3d_oracle_rig_from_transforms(int32_t n_cam, const double *T_base_to_cam, float *Rt_out, float *F_out);

/* HUN:60-397 restated with index arrays; column-major dist[row + nRows*col]. */
int se3d_oracle_hungarian(int32_t *assignment, double *cost, const double *dist, int32_t n_rows, int32_t n_cols);

/* S3D:440-465 (+425-438): pts = [n][3] (x, y, conf) normalised, Rt = [n][12]. */
int se3d_oracle_dlt(int32_t n, const float *Rt, const float *pts, int32_t weight_by_conf, float *X3, double *reproj);

/* S3D:471-523: cov2 = [n][4] row-major 2x2, mean = weighted triangulation result; cov_out = 9 floats row-major. */
int se3d_oracle_ut_cov(int32_t n, const float *Rt, const float *pts, const float *cov2, const float *mean3,
                       float *cov_out9);

/* S3D:335-390 for one (hypothesis-view, detection) pair: mean over joints of d1+d2 and #joints used. */
double se3d_oracle_pair_cost(const float *F9, const float *kp_a /*[17][3]*/, const float *kp_b /*[17][3]*/,
                             int32_t *n_joints_used);

/* S3D:525-997.  omp_threads <= 1: serial (canonical order).  > 1: OpenMP over hypotheses like S3D:676. */
int se3d_oracle_triangulate(const se3d_rig *rig, const se3d_params *params, const se3d_frame_in *frame,
                            se3d_frame_out *out, int32_t omp_threads);

/* pose_prior node state + callback (PP:505-921). */
typedef struct se3d_oracle_pp se3d_oracle_pp;
se3d_oracle_pp *se3d_oracle_pp_create(const se3d_params *params);
void se3d_oracle_pp_destroy(se3d_oracle_pp *pp);
int se3d_oracle_pp_step(se3d_oracle_pp *pp, int32_t n_det, const se3d_person3d *dets, int32_t n_cam,
                        const float *fb_delay_per_cam, double t, se3d_fused_out *out, int32_t omp_threads);

/* LM + marginals on one explicit graph (rows F, R) for known-answer tests.
 * K variables (3-D points); unary: key, m[3], cov[9]; range: k1, k2, L, sigma. */
int se3d_oracle_lm(int32_t K, const double *init /*[K][3]*/, int32_t n_unary, const int32_t *u_key,
                   const double *u_meas, const double *u_cov, int32_t n_range, const int32_t *r_k1,
                   const int32_t *r_k2, const double *r_len, const double *r_sigma, double *result /*[K][3]*/,
                   double *marg_cov /*[K][9] or NULL*/, int32_t *iterations, double *final_error);

/* pose_reprojection fusedSkeletonCallback (skeleton_reproj_mult_node.cpp:139-235), all double precision. */
int se3d_oracle_reproject(int32_t n_cam, const double *T_base_to_cam, const double *P_fx_fy_cx_cy, const int32_t *wh, int32_t pose_method,
                          int32_t n_persons, const se3d_person3d *persons, se3d_reproj_out *out);

#ifdef __cplusplus
}
#endif
#endif
