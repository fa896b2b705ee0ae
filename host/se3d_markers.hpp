// se3d_markers.hpp — the rviz MarkerArray outputs of the two nodes, built on the host from the persons the accelerated path returns.
//   skeleton_markers  <->  the "joints" / "joint_spheres" markers of triangulate_persons      skeleton_3d_triang_mult_node.cpp:689-716, 858-921, 967-971
//   fused_markers     <->  the "skeleton_fused" / "joints_fused" markers of skeletonCallback   pose_prior_mult_node.cpp:598-627, 273-375, 833-852
//   FusedMarkerState  <->  the DELETE markers of remove_old_tracks / the track merge           pose_prior_mult_node.cpp:191-209, 878-891
// Plain C++ (no ROS, no GPU): the Marker struct mirrors visualization_msgs/Marker field for field, so a catkin wrapper copies it out.
// Deviations from the reference, all confined to the picture in rviz (documented in INTEGRATION.md):
//   * the stage-1 markers are built from the PUBLISHED persons; the reference builds them inside the hypothesis loop, before the
//     root-distance reset (S3D:936-951), so a joint that filter removed is still drawn there, and it numbers them by hypothesis index
//     (here: index in the published list); the `vis_cov` ellipsoids (off by default, S3D:77; covariance_markers below) are drawn from the
//     PUBLISHED covariances: stage 1 after the limb-length inflation (the reference draws before S3D:861-883 inflates), stage 2 with the
//     mid hip's x 100^2 rescale undone as the reference draws before it (PP:795-814); eigenvector signs are free, so a quaternion may
//     differ from Eigen's by a half turn about an axis of the same ellipsoid;
//   * DELETE markers are derived from the published ids (an id not seen for t_max_unobserved), the reference emits them from its track list.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <map>
#include <string>
#include <vector>

#include "person_msgs.hpp"

namespace visualization_msgs {

struct ColorRGBA { float r = 0, g = 0, b = 0, a = 0; };
struct Marker {  // visualization_msgs/Marker
  enum : int32_t { SPHERE = 2, LINE_LIST = 5, SPHERE_LIST = 7, ADD = 0, DELETE = 2 };
  person_msgs::Header header;
  std::string ns;
  int32_t id = 0, type = 0, action = ADD;
  person_msgs::Point position;                           // pose.position
  std::array<double, 4> orientation{{0, 0, 0, 1}};       // pose.orientation x, y, z, w
  std::array<double, 3> scale{{0, 0, 0}};
  ColorRGBA color;
  double lifetime = 0.0;                                 // seconds; 0 = forever
  std::vector<person_msgs::Point> points;
  std::vector<ColorRGBA> colors;
};
struct MarkerArray { std::vector<Marker> markers; };

}  // namespace visualization_msgs

namespace se3d {

// fusion body parts (person_msgs/include/person_msgs/fusion_body_parts.h)
enum FusionPart { FU_Nose = 0, FU_Neck, FU_RShoulder, FU_RElbow, FU_RWrist, FU_LShoulder, FU_LElbow, FU_LWrist, FU_MidHip, FU_RHip, FU_RKnee,
                  FU_RAnkle, FU_LHip, FU_LKnee, FU_LAnkle, FU_REye, FU_LEye, FU_REar, FU_LEar, FU_Head, FU_Belly, FU_NUM };

// the joint colours of both nodes (S3D:1141-1170 = PP:150-178), by fusion body part, 8-bit components
inline visualization_msgs::ColorRGBA joint_color(int fusion_idx) {
  static const unsigned char rgb[29][3] = {
      {255, 0, 0},   {85, 170, 0},  {0, 255, 0},   {0, 255, 170}, {0, 170, 255}, {85, 255, 0},  {0, 255, 85},  {0, 255, 255}, {0, 85, 170},  {0, 0, 255},
      {100, 0, 255}, {255, 0, 255}, {0, 85, 255},  {50, 0, 255},  {170, 0, 255}, {255, 170, 0}, {255, 85, 0},  {170, 255, 0}, {255, 255, 0}, {255, 150, 0},
      {42, 128, 85}, {50, 0, 255},  {100, 0, 255}, {150, 0, 255}, {200, 0, 255}, {255, 0, 200}, {255, 0, 150}, {255, 0, 100}, {255, 0, 50}};
  visualization_msgs::ColorRGBA c;
  c.r = rgb[fusion_idx][0] / 255.f; c.g = rgb[fusion_idx][1] / 255.f; c.b = rgb[fusion_idx][2] / 255.f; c.a = 1.f;
  if (fusion_idx == FU_Belly) c.g = 0.5f;  // the reference writes 0.5, not 128 / 255 (S3D:1162)
  return c;
}

namespace detail {
inline visualization_msgs::Marker list_marker(const person_msgs::Header& h, int32_t type, const char* ns, int32_t id, double lifetime) {
  visualization_msgs::Marker m;
  m.header = h; m.type = type; m.ns = ns; m.id = id; m.lifetime = lifetime;
  if (type == visualization_msgs::Marker::LINE_LIST) { m.scale = {{0.05, 0, 0}}; m.color.r = 1.f; m.color.a = 1.f; }
  else { m.scale = {{0.07, 0.07, 0.07}}; m.color.r = m.color.g = 0.5f; m.color.a = 1.f; }
  return m;
}
// one segment of a LINE_LIST: from the already drawn end point of `from` (the joint itself when it was not drawn), coloured like the joint
inline void add_segment(visualization_msgs::Marker& skel, std::vector<int>& msg_idx, int from, const person_msgs::Point& joint,
                        const visualization_msgs::ColorRGBA& col) {
  if (from >= 0 && msg_idx[from] != -1 && msg_idx[from] < (int)skel.points.size()) skel.points.push_back(skel.points[msg_idx[from]]);
  else skel.points.push_back(joint);
  skel.points.push_back(joint);
  skel.colors.push_back(col); skel.colors.push_back(col);
}
}  // namespace detail

// Stage 1: two markers per published person, in list order (S3D:967-971): LINE_LIST "joints" then SPHERE_LIST "joint_spheres", lifetime 2 s.
// pose_method "simple" | "h36m" selects the input keypoint order, its parent table and its map to the fusion body parts (S3D:100, 130, 1099-1112).
inline void skeleton_markers(const person_msgs::PersonCovList& persons3d, const std::string& pose_method, visualization_msgs::MarkerArray& out) {
  static const int parent_simple[17] = {-1, 0, 0, 1, 2, 0, 0, 5, 6, 7, 8, 5, 6, 11, 12, 13, 14};
  static const int parent_h36m[17] = {-1, 0, 0, 2, 3, 2, 2, 5, 6, 7, 8, 4, 4, 11, 12, 13, 14};
  static const int fusion_simple[17] = {FU_Nose, FU_LEye, FU_REye, FU_LEar, FU_REar, FU_LShoulder, FU_RShoulder, FU_LElbow, FU_RElbow, FU_LWrist, FU_RWrist,
                                        FU_LHip, FU_RHip, FU_LKnee, FU_RKnee, FU_LAnkle, FU_RAnkle};
  static const int fusion_h36m[17] = {FU_Nose, FU_Head, FU_Neck, FU_Belly, FU_MidHip, FU_LShoulder, FU_RShoulder, FU_LElbow, FU_RElbow, FU_LWrist, FU_RWrist,
                                      FU_LHip, FU_RHip, FU_LKnee, FU_RKnee, FU_LAnkle, FU_RAnkle};
  const bool h36m = pose_method == "h36m";
  const int* parent = h36m ? parent_h36m : parent_simple;
  const int* fusion = h36m ? fusion_h36m : fusion_simple;
  for (size_t p = 0; p < persons3d.persons.size(); ++p) {
    const auto& person = persons3d.persons[p];
    auto skel = detail::list_marker(persons3d.header, visualization_msgs::Marker::LINE_LIST, "joints", (int32_t)p, 2.0);
    auto joints = detail::list_marker(persons3d.header, visualization_msgs::Marker::SPHERE_LIST, "joint_spheres", (int32_t)p, 2.0);
    std::vector<int> msg_idx(17, -1);
    for (int k = 0; k < 17; ++k) {  // S3D:858-921
      if ((int)person.keypoints.size() <= fusion[k]) continue;
      const auto& kp = person.keypoints[fusion[k]];
      if (kp.score <= 0) continue;
      const auto col = joint_color(fusion[k]);
      joints.points.push_back(kp.joint);
      joints.colors.push_back(col);
      detail::add_segment(skel, msg_idx, parent[k], kp.joint, col);
      msg_idx[k] = (int)skel.points.size() - 1;
    }
    out.markers.push_back(std::move(skel));
    out.markers.push_back(std::move(joints));
  }
}

// Stage 2: per published fused person SPHERE_LIST "joints_fused" then LINE_LIST "skeleton_fused" (PP:849-852), id = track id, no lifetime.
// The segment of a joint starts at the first of its candidate neighbours that has been drawn (addJointToSkeleton, PP:273-375).
inline void fused_markers(const person_msgs::PersonCovList& fused, visualization_msgs::MarkerArray& out) {
  for (const auto& person : fused.persons) {
    auto skel = detail::list_marker(fused.header, visualization_msgs::Marker::LINE_LIST, "skeleton_fused", (int32_t)person.id, 0.0);
    auto joints = detail::list_marker(fused.header, visualization_msgs::Marker::SPHERE_LIST, "joints_fused", (int32_t)person.id, 0.0);
    std::vector<int> msg_idx(FU_NUM, -1);
    auto drawn = [&](int k) { return k >= 0 && msg_idx[k] != -1 && msg_idx[k] < (int)skel.points.size(); };
    auto first_drawn = [&](std::initializer_list<int> cands) { for (int k : cands) if (drawn(k)) return k; return -1; };
    for (int k = 0; k < FU_NUM && k < (int)person.keypoints.size(); ++k) {
      const auto& kp = person.keypoints[k];
      if (kp.score <= 0) continue;   // the reference walks the estimated values only: every published joint has score >= 0.10 (PP:779-784)
      const auto col = joint_color(k);
      joints.points.push_back(kp.joint);
      joints.colors.push_back(col);
      bool known = true;
      int from = -1;
      switch (k) {
        case FU_Nose: from = -1; break;
        case FU_Head: from = first_drawn({FU_Nose}); break;
        case FU_RElbow: case FU_RWrist: case FU_LElbow: case FU_LWrist: case FU_RKnee: case FU_RAnkle: case FU_LKnee: case FU_LAnkle:
          from = first_drawn({k - 1}); break;
        case FU_RShoulder: case FU_LShoulder: case FU_MidHip: from = first_drawn({FU_Neck, FU_Nose}); break;
        case FU_Belly:  // two segments: to the neck and to the mid hip
          detail::add_segment(skel, msg_idx, first_drawn({FU_Neck}), kp.joint, col);
          from = first_drawn({FU_MidHip}); break;
        case FU_RHip: case FU_LHip: from = first_drawn({FU_MidHip, FU_Neck, k - 7}); break;
        case FU_REye: case FU_LEye: case FU_Neck: from = first_drawn({FU_Nose}); break;
        case FU_REar: case FU_LEar: from = first_drawn({k - 2}); break;
        default: known = false; break;
      }
      if (!known) continue;
      detail::add_segment(skel, msg_idx, from, kp.joint, col);
      msg_idx[k] = (int)skel.points.size() - 1;
    }
    if (joints.points.empty()) continue;
    out.markers.push_back(std::move(joints));
    out.markers.push_back(std::move(skel));
  }
}

// ---- `vis_cov`: one SPHERE marker per joint drawn as the 95 % ellipsoid of its 3x3 covariance (setMarkerPose, S3D:279-309 = PP:237-254) ----
// Eigen-decomposition of a symmetric 3x3 matrix by cyclic Jacobi rotations: eigenvalues ascending (the order of Eigen's
// SelfAdjointEigenSolver, which the reference reads as scale x, y, z), eigenvectors in the columns of V.
inline void sym3_eigen(const double c6[6], double eval[3], double V[3][3]) {
  double A[3][3] = {{c6[0], c6[1], c6[2]}, {c6[1], c6[3], c6[4]}, {c6[2], c6[4], c6[5]}};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) V[i][j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 50; ++sweep) {
    const double off = std::fabs(A[0][1]) + std::fabs(A[0][2]) + std::fabs(A[1][2]);
    const double diag = std::fabs(A[0][0]) + std::fabs(A[1][1]) + std::fabs(A[2][2]);
    if (off <= 1e-18 * diag || off == 0.0) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (A[p][q] == 0.0) continue;
        const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
        const double c = 1.0 / std::sqrt(t * t + 1.0), s_ = t * c;
        for (int k = 0; k < 3; ++k) { const double x = A[k][p], y = A[k][q]; A[k][p] = c * x - s_ * y; A[k][q] = s_ * x + c * y; }
        for (int k = 0; k < 3; ++k) { const double x = A[p][k], y = A[q][k]; A[p][k] = c * x - s_ * y; A[q][k] = s_ * x + c * y; }
        for (int k = 0; k < 3; ++k) { const double x = V[k][p], y = V[k][q]; V[k][p] = c * x - s_ * y; V[k][q] = s_ * x + c * y; }
      }
  }
  int idx[3] = {0, 1, 2};
  for (int i = 0; i < 2; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (A[idx[j]][idx[j]] < A[idx[i]][idx[i]]) std::swap(idx[i], idx[j]);
  double Vs[3][3];
  for (int k = 0; k < 3; ++k) { eval[k] = A[idx[k]][idx[k]]; for (int r = 0; r < 3; ++r) Vs[r][k] = V[r][idx[k]]; }
  for (int r = 0; r < 3; ++r) for (int k = 0; k < 3; ++k) V[r][k] = Vs[r][k];
}

// pose (position, orientation of the principal axes, a proper rotation) and scale (2 x 2.7955 sigma per axis: 2.7955 = sqrt(chi2inv(0.95, 3)))
inline void set_marker_pose(visualization_msgs::Marker& m, const person_msgs::Point& joint, const double c6[6], double eval_out[3] = nullptr) {
  double ev[3], V[3][3];
  sym3_eigen(c6, ev, V);
  const double det = V[0][0] * (V[1][1] * V[2][2] - V[1][2] * V[2][1]) - V[0][1] * (V[1][0] * V[2][2] - V[1][2] * V[2][0]) + V[0][2] * (V[1][0] * V[2][1] - V[1][1] * V[2][0]);
  if (!(det > 0.0))   // "Determinant must be +1!" (S3D:289-292)
    for (auto& row : V) for (double& v : row) v = -v;
  // rotation matrix -> quaternion (the trace method Eigen::Quaterniond(Matrix3d) uses)
  double q[4];  // w, x, y, z
  const double tr = V[0][0] + V[1][1] + V[2][2];
  if (tr > 0.0) {
    double t = std::sqrt(tr + 1.0);
    q[0] = 0.5 * t; t = 0.5 / t;
    q[1] = (V[2][1] - V[1][2]) * t; q[2] = (V[0][2] - V[2][0]) * t; q[3] = (V[1][0] - V[0][1]) * t;
  } else {
    int i = 0;
    if (V[1][1] > V[0][0]) i = 1;
    if (V[2][2] > V[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double t = std::sqrt(V[i][i] - V[j][j] - V[k][k] + 1.0);
    q[1 + i] = 0.5 * t; t = 0.5 / t;
    q[0] = (V[k][j] - V[j][k]) * t; q[1 + j] = (V[j][i] + V[i][j]) * t; q[1 + k] = (V[k][i] + V[i][k]) * t;
  }
  m.position = joint;
  m.orientation = {{q[1], q[2], q[3], q[0]}};
  for (int a = 0; a < 3; ++a) m.scale[a] = 2.0 * 2.7955 * std::sqrt(ev[a]);
  if (eval_out) for (int a = 0; a < 3; ++a) eval_out[a] = ev[a];
}

// running minimum / maximum of the covariance eigenvalues, printed as "Sigmas 3D [mm]" when the node exits (S3D:303-309, 1243-1247)
struct SigmaRange {
  double min_ev[3] = {1e30, 1e30, 1e30}, max_ev[3] = {0, 0, 0};
  void add(const double ev[3]) { for (int a = 0; a < 3; ++a) { min_ev[a] = std::min(min_ev[a], ev[a]); max_ev[a] = std::max(max_ev[a], ev[a]); } }
  double min_mm(int a) const { return std::sqrt(min_ev[a]) * 1000.0; }
  double max_mm(int a) const { return std::sqrt(max_ev[a]) * 1000.0; }
};

// Stage 1, vis_cov = true: "joint_cov_3d" SPHERE markers, lifetime 5 s, id = 21 * person + input keypoint index, for the keypoints whose fusion
// part is below 15 (no eyes / ears / head / belly), colour of the joint at alpha 0.5 (S3D:887-898).
inline void covariance_markers_3d(const person_msgs::PersonCovList& persons3d, const std::string& pose_method, visualization_msgs::MarkerArray& out,
                                  SigmaRange* range = nullptr) {
  static const int fusion_simple[17] = {FU_Nose, FU_LEye, FU_REye, FU_LEar, FU_REar, FU_LShoulder, FU_RShoulder, FU_LElbow, FU_RElbow, FU_LWrist, FU_RWrist,
                                        FU_LHip, FU_RHip, FU_LKnee, FU_RKnee, FU_LAnkle, FU_RAnkle};
  static const int fusion_h36m[17] = {FU_Nose, FU_Head, FU_Neck, FU_Belly, FU_MidHip, FU_LShoulder, FU_RShoulder, FU_LElbow, FU_RElbow, FU_LWrist, FU_RWrist,
                                      FU_LHip, FU_RHip, FU_LKnee, FU_RKnee, FU_LAnkle, FU_RAnkle};
  const int* fusion = pose_method == "h36m" ? fusion_h36m : fusion_simple;
  for (size_t p = 0; p < persons3d.persons.size(); ++p)
    for (int k = 0; k < 17; ++k) {
      const int f = fusion[k];
      if (f >= 15 || (int)persons3d.persons[p].keypoints.size() <= f) continue;
      const auto& kp = persons3d.persons[p].keypoints[f];
      if (kp.score <= 0) continue;
      visualization_msgs::Marker m;
      m.header = persons3d.header; m.lifetime = 5.0; m.type = visualization_msgs::Marker::SPHERE; m.ns = "joint_cov_3d";
      m.id = (int32_t)(FU_NUM * p + k);
      m.color = joint_color(f); m.color.a = 0.50f;
      double ev[3];
      set_marker_pose(m, kp.joint, kp.cov.data(), ev);
      if (range) range->add(ev);
      out.markers.push_back(std::move(m));
    }
}

// Stage 2, vis_cov = true: "joint_cov_fused" SPHERE markers, lifetime g_t_max_unobserved, id = 21 * person + fusion part, parts below 15
// (PP:795-807).  The reference draws the mid hip's marginal before it is rescaled by g_root_sigma_factor^2 (PP:813-814): undone here.
inline void covariance_markers_fused(const person_msgs::PersonCovList& fused, visualization_msgs::MarkerArray& out, double t_max_unobserved = 1.0) {
  for (size_t p = 0; p < fused.persons.size(); ++p)
    for (int k = 0; k < 15 && k < (int)fused.persons[p].keypoints.size(); ++k) {
      const auto& kp = fused.persons[p].keypoints[k];
      if (kp.score <= 0) continue;
      double c6[6];
      for (int a = 0; a < 6; ++a) c6[a] = k == FU_MidHip ? kp.cov[a] / (100.0 * 100.0) : kp.cov[a];
      visualization_msgs::Marker m;
      m.header = fused.header; m.lifetime = t_max_unobserved; m.type = visualization_msgs::Marker::SPHERE; m.ns = "joint_cov_fused";
      m.id = (int32_t)(FU_NUM * p + k);
      m.color = joint_color(k); m.color.a = 0.50f;
      set_marker_pose(m, kp.joint, c6);
      out.markers.push_back(std::move(m));
    }
}

// DELETE markers for tracks that are gone (PP:191-209; after a merge PP:878-891): an id that was published and has not been seen for
// t_max_unobserved seconds.  update() appends them to `out` and forgets the id.
class FusedMarkerState {
 public:
  explicit FusedMarkerState(double t_max_unobserved = 1.0) : t_max_(t_max_unobserved) {}
  void update(const person_msgs::PersonCovList& fused, visualization_msgs::MarkerArray& out) {
    const double t = fused.header.stamp;
    for (const auto& p : fused.persons) last_seen_[p.id] = t;
    for (auto it = last_seen_.begin(); it != last_seen_.end();) {
      if (t - it->second > t_max_) {
        for (const char* ns : {"skeleton_fused", "joints_fused"}) {
          visualization_msgs::Marker del;
          del.header = fused.header; del.ns = ns; del.id = (int32_t)it->first; del.action = visualization_msgs::Marker::DELETE;
          out.markers.push_back(std::move(del));
        }
        it = last_seen_.erase(it);
      } else ++it;
    }
  }
  void reset() { last_seen_.clear(); }

 private:
  double t_max_;
  std::map<uint32_t, double> last_seen_;
};

}  // namespace se3d
