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
 *   - Constants (row K, the part that lives in a stand-alone header): PINNED — FUSION_BODY_PARTS joint indices and vel_sigmas against the
 *     reference's fusion_body_parts.h compiled in place (oracle/fbp_ref_shim.cpp -> oracle/_ref/libfbp_ref.so,
 *     tests/golden/fusion_body_parts.json, tests/test_oracle_fbp_golden.py).  The named node constants and the pose-specific tables sit inside
 *     the ROS node sources: read out BY NAME into tests/golden/pose_constants.json and compared with this library's values by the same test.
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

/* Config 2 of SURVEY 8d — batched DLT with the association GIVEN: rows N (S3D:312-333), T (:440-465), P (:425-438), O (:748-844),
 * U (:471-523) per hypothesis and joint, scattered into the 21-slot fusion layout (:849-857); no sweep, no post-processing, no merge.
 * assoc: [n_hyp][n_cam] index into that camera's persons[] or -1; persons_out: [n_hyp]. */
int se3d_oracle_triangulate_assoc(const se3d_rig *rig, const se3d_params *params, const se3d_frame_in *frame, int32_t n_hyp,
                                  const int32_t *assoc, se3d_person3d *persons_out);

/* Branch counters: how often each data-dependent branch of the reference fired since the last reset (process-wide).  The branch
 * tests (tests/test_gpu_branches.py, tests/test_oracle_branches.py) assert > 0 so that a green parity test proves the branch ran. */
enum {
  SE3D_ORACLE_STAT_SWEEP_HUNGARIAN = 0,  /* ambiguous mask -> Hungarian on the unmasked C (S3D:628-634) */
  SE3D_ORACLE_STAT_SPAWN_UNMASKED,       /* assigned but not feasible -> detection spawns a hypothesis (S3D:640-648) */
  SE3D_ORACLE_STAT_OUTLIER3_REMOVED,     /* n == 3: camera dropped by the pairwise epipolar test (S3D:748-792) */
  SE3D_ORACLE_STAT_OUTLIER_LOO_REMOVED,  /* n >= 4: leave-one-out re-triangulation kept (S3D:793-838) */
  SE3D_ORACLE_STAT_SCORE_DOWNWEIGHTED,   /* still-large reprojection error (S3D:840-844) */
  SE3D_ORACLE_STAT_NAN_COVARIANCE,       /* non-SPD 2-D covariance -> NaN 3-D covariance (S3D:473-475) */
  SE3D_ORACLE_STAT_ROOT_DIST_RESET,      /* joint > 2 m from the root reset (S3D:937-953) */
  SE3D_ORACLE_STAT_FEET_REJECT,          /* |feet height| > 0.5 m (S3D:955-966) */
  SE3D_ORACLE_STAT_PERSON_REJECTED,      /* <= 9 valid joints left (S3D:968) */
  SE3D_ORACLE_STAT_DUPLICATE_MERGE,      /* persons closer than 0.20 m merged (S3D:984-996) */
  SE3D_ORACLE_STAT_EMPTY_FRAME,          /* pose_prior: no detections (PP:537-546) */
  SE3D_ORACLE_STAT_TRACK_ASSOC_REJECTED, /* assignment with cost > 5.0 undone (PP:562-565) */
  SE3D_ORACLE_STAT_TRACK_TIMEOUT,        /* track unobserved > 1 s removed (PP:191-211) */
  SE3D_ORACLE_STAT_JOINT_VANISHED,       /* prevEstimate key erased, velocity buffer zeroed (PP:483-503) */
  SE3D_ORACLE_STAT_LM_FAILED,            /* optimisation failed -> measurements (PP:751-758) */
  SE3D_ORACLE_STAT_MARGINALS_FAILED,     /* indeterminate marginals -> 0.10^2 I (PP:788-793) */
  SE3D_ORACLE_STAT_TRACK_MERGE,          /* tracks closer than 0.20 m merged (PP:870-903) */
  SE3D_ORACLE_STAT_ID_REWRITE,           /* published id rewritten to the surviving track's (PP:892-899) */
  SE3D_ORACLE_STAT_COUNT
};
void se3d_oracle_stats_reset(void);
void se3d_oracle_stats_get(int64_t *out /* [SE3D_ORACLE_STAT_COUNT] */);
/* the restated FUSION_BODY_PARTS constants: joint indices in declaration order + NUM_KEYPOINTS, vel_sigmas (FBP:3-24, 33) */
void se3d_oracle_fusion_constants(int32_t *indices22, double *vel_sigmas21);
/* the restated node constants (order documented in se3d_oracle.cpp / oracle_py.NODE_CONSTANT_NAMES) and per-pose-method tables */
void se3d_oracle_node_constants(double *scalars16);
void se3d_oracle_bone_table(int32_t norm_height, int32_t *k1, int32_t *k2, double *length, double *sigma, int32_t *not_belly);   /* 21 entries each */
void se3d_oracle_pose_tables(int32_t pose_method, int32_t *kp_parent17, double *limb_length17, double *limb_sigma17, int32_t *kp2fusion17,
                             double *shoulder2);

/* pose_prior node state + callback (PP:505-921). */
typedef struct se3d_oracle_pp se3d_oracle_pp;
se3d_oracle_pp *se3d_oracle_pp_create(const se3d_params *params);
void se3d_oracle_pp_destroy(se3d_oracle_pp *pp);
int se3d_oracle_pp_step(se3d_oracle_pp *pp, int32_t n_det, const se3d_person3d *dets, int32_t n_cam,
                        const float *fb_delay_per_cam, double t, se3d_fused_out *out, int32_t omp_threads);
/* test bookkeeping of the last step: the track id of every detection (PP:570-580); the (removed id, surviving id) pairs of its merges (PP:870-903) */
int32_t se3d_oracle_pp_last_det_ids(const se3d_oracle_pp *pp, int32_t *ids, int32_t capacity);
int32_t se3d_oracle_pp_last_merges(const se3d_oracle_pp *pp, int32_t *pairs, int32_t capacity_pairs);
int32_t se3d_oracle_pp_tracks(const se3d_oracle_pp *pp, int32_t *id_obs, double *state70, int32_t capacity);   /* live tracks: (id, num_obs); t_prev, height_prev, root_prev[3], joint mask, prevEstimate[21][3] */

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
