// se3d_nodes.hpp — C++ host layer above the C-ABI (include/se3d.h), mirroring the reference nodes' internal interfaces:
//   Skeleton3DNode::triangulate_persons   <->  triangulate_persons(...)            skeleton_3d_triang_mult_node.cpp:525
//   Skeleton3DNode::skeletonThreadStep    <->  body of skeletonThreadCallback      skeleton_3d_triang_mult_node.cpp:1029-1084
//   PosePriorNode::skeletonCallback       <->  skeletonCallback(PersonCovList)     pose_prior_mult_node.cpp:505
//   ReprojectionNode::fusedSkeletonCallback <-> fusedSkeletonCallback(PersonCovList) skeleton_reproj_mult_node.cpp:139
//   Skeleton3DFrontEnd                    <->  the approximate-time synchroniser in front of skeletonCallback   S3D:1218-1223
// ROS is not available in the build image, so messages are the plain mirrors of host/person_msgs.hpp; a catkin wrapper
// only has to forward its callbacks to these methods.  All arithmetic runs in libse3d_hip.so on the GPU.
#pragma once
#include <algorithm>
#include <cmath>
#include <functional>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/se3d.h"
#include "person_msgs.hpp"
#include "se3d_sync.hpp"

namespace se3d {

class HypothesisShard;   // se3d_exchange.hpp: the hypotheses / detections of ONE stream split over the ranks of a node (north_star config 5)

struct NodeParams {                 // ROS private parameters (S3D:1095-1097, PP:930-932)
  std::string pose_method = "simple";
  double max_epi_dist = 0.050;
  bool norm_height = false;
};

struct CameraCalib {
  std::array<double, 16> T_base_to_cam;  // row-major 4x4: lookupTransform(cam, base) (S3D:166)
  double fx, fy, cx, cy;                 // CameraInfo K (S3D:226)
  int width = 640, height = 480;         // CameraInfo width / height (REPROJ:161-162)
};

// Owns one se3d_ctx (one GPU stream, stage-1 workspace, stage-2 track state).  The library works inside fixed limits (detections per
// camera, hypotheses per frame, live tracks) and reports when a frame exceeds them (SE3D_ERR_CAPACITY, SE3D_FLAG_HYP_OVERFLOW,
// SE3D_S2FLAG_TRACK_OVERFLOW); the reference has no such limits — its vectors grow (S3D:636-673 H.push_back, PP:570-580
// g_tracks.push_back).  The nodes below therefore never publish a truncated result: they call grow(), which creates a context with
// larger limits, moves the pose_prior state over (se3d_refine_save / _load) and re-applies the reprojection cameras, and run the frame
// again.  The only hard bound left is the library's SE3D capacity of 256 detections / hypotheses / tracks (kCapacityMax): beyond it the
// node throws and names the limit.
class Context {
 public:
  static constexpr int kCapacityMax = 256;   // SE3D_MUNKRES_MAX of the library: rows / columns of the device Munkres
  // options: se3d_options of the library (include/se3d.h; nullptr = defaults); kept and applied again when the context grows
  // max_frames: frames a resident batch may hold (1: the nodes' operating point, one synchronised frame per callback; larger for the resident calls
  // se3d_stage_frames / se3d_run_stage1 / se3d_run_stage2_walk on get(), e.g. a replayed walk exchanged by se3d::SkeletonExchange)
  Context(const std::vector<CameraCalib>& cams, const NodeParams& prm, int max_det_per_cam = 16, int max_hyp = 64, int device = 0, const se3d_options* options = nullptr,
          int max_frames = 1);
  ~Context();
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  se3d_ctx* get() const { return ctx_; }
  int n_cam() const { return n_cam_; }
  int max_hyp() const { return max_hyp_; }            // = limits.max_hyp = limits.max_tracks
  int max_det_per_cam() const { return max_det_; }
  int n_grown() const { return n_grown_; }            // how often the limits were raised (diagnostic)
  const std::vector<float>& fundamental() const { return F_; }
  const std::vector<CameraCalib>& cameras() const { return cams_; }
  // Raise the limits to at least (min_det_per_cam, min_hyp) — doubling, so that a growing scene costs O(log) re-creations — keeping the
  // pose_prior state.  Throws std::runtime_error naming the limit when kCapacityMax is not enough.
  void grow(int min_det_per_cam, int min_hyp, const char* why);
  void set_reprojection_cameras(const std::vector<double>& T, const std::vector<double>& P, const std::vector<int32_t>& wh);

 private:
  se3d_ctx* create(int max_det, int max_hyp) const;
  se3d_ctx* ctx_ = nullptr;
  int n_cam_ = 0, max_hyp_ = 0, max_det_ = 0, device_ = 0, n_grown_ = 0, max_frames_ = 1;
  NodeParams prm_;
  se3d_options opt_{};
  std::vector<float> Rt_, K_, F_;
  std::vector<CameraCalib> cams_;
  std::vector<double> rp_T_, rp_P_;      // reprojection cameras, re-applied after grow()
  std::vector<int32_t> rp_wh_;
};

// One synchronised set of per-camera messages as the C-ABI takes it (se3d_frame_in over storage owned by this object): what
// Skeleton3DNode::triangulate_persons builds per callback, exposed for callers of the resident entry points (se3d_stage_frames).
struct FlatFrame {
  std::vector<int32_t> off;
  std::vector<se3d_person2d> flat;
  std::vector<float> fb;
  se3d_frame_in in{};
};
void flatten_frame(const std::vector<person_msgs::Person2DList>& people, double stamp, FlatFrame& out);

class Skeleton3DNode {
 public:
  explicit Skeleton3DNode(std::shared_ptr<Context> ctx) : ctx_(std::move(ctx)) {}
  // S3D:525 — fills persons3d_msg.persons (21 keypoints each); assoc (optional) receives [n_hyp][n_cam] detection indices
  void triangulate_persons(const std::vector<person_msgs::Person2DList>& people, person_msgs::PersonCovList& persons3d_msg,
                           std::vector<int32_t>* assoc = nullptr, int* n_hyp = nullptr);
  // S3D:1029-1084 — pivot selection, non-monotonic drop, stale-view replacement (> 67 ms), header bookkeeping, timing buckets.
  // Returns false when the frame is skipped (delta_t <= 0, S3D:1043-1046).
  bool skeletonThreadStep(std::vector<person_msgs::Person2DList> people, person_msgs::PersonCovList& persons3d_msg);
  // "avg runtime" report of S3D:1234-1241: index 0 = all frames, n = frames with n output persons (n < 10)
  const std::vector<double>& timings_ms() const { return timings_; }
  const std::vector<int>& timing_counts() const { return timing_cnt_; }
  // Multi-GPU latency mode (one process per GPU, every rank fed the same messages): triangulate_persons runs the resident shard calls —
  // association on every rank, the own hypotheses' triangulation, one all-gather, post-processing on every rank — instead of se3d_triangulate;
  // every rank publishes the single-rank result bit for bit.  The limits are fixed in this mode (an overflow throws instead of growing).
  void set_shard(HypothesisShard* shard) { shard_ = shard; }

 private:
  std::shared_ptr<Context> ctx_;
  HypothesisShard* shard_ = nullptr;
  double last_stamp_ = 0.0;
  std::vector<double> timings_ = std::vector<double>(10, 0.0);
  std::vector<int> timing_cnt_ = std::vector<int>(10, 0);
};

// The N per-camera Person2DList subscriptions of main() (S3D:1140-1160, 1218-1223): messages go through the approximate-time
// synchroniser with the node's parameters (queue max(3, 1 + N/4), 20 ms lower bound, age penalty 2.0); every published tuple is
// handed to `on_tuple` (the reference's skeletonCallback stores it for the worker thread, S3D:999-1027).
class Skeleton3DFrontEnd {
 public:
  using Tuple = std::vector<std::shared_ptr<const person_msgs::Person2DList>>;
  Skeleton3DFrontEnd(int num_cameras, std::function<void(const Tuple&)> on_tuple)
      : sync_(std::max(3u, 1u + (unsigned)num_cameras / 4u), num_cameras,
              [](const person_msgs::Person2DList& m) { return (int64_t)std::llround(m.header.stamp * 1e9); }) {
    sync_.setInterMessageLowerBound(20000000);
    sync_.setAgePenalty(2.0);
    sync_.registerCallback(std::move(on_tuple));
  }
  void add(std::shared_ptr<const person_msgs::Person2DList> msg, int cam) { sync_.add(std::move(msg), cam); }

 private:
  ApproximateTimeSync<person_msgs::Person2DList> sync_;
};

class PosePriorNode {
 public:
  explicit PosePriorNode(std::shared_ptr<Context> ctx, int seq = 0) : ctx_(std::move(ctx)), seq_(seq) {}
  void reset();  // PP:182-189
  // PP:505 — consumes persons_3d, produces persons3d_fused and persons3d_fused_pred (headers / ts_per_cam copied, fb_delay_per_cam
  // replaced by the predicted latency, PP:528-532)
  void skeletonCallback(const person_msgs::PersonCovList& persons, person_msgs::PersonCovList& persons_fused,
                        person_msgs::PersonCovList& persons_fused_pred);
  // Multi-GPU latency mode: the callback refines the stage-1 persons RESIDENT in the context (frame slot 0: what the sharded
  // Skeleton3DNode::triangulate_persons of the same context has just produced — `persons` carries the same list) with the detections split over
  // the ranks: se3d_run_stage2_shard_pre -> all-gather -> _post.
  void set_shard(HypothesisShard* shard) { shard_ = shard; }

 private:
  std::shared_ptr<Context> ctx_;
  HypothesisShard* shard_ = nullptr;
  int seq_;
  int n_tracks_ = 0;   // live tracks after the last callback: a step adds at most one track per detection, so n_tracks_ + n_det is a bound
};

class ReprojectionNode {
 public:
  // uploads the double-precision camera poses / projection matrices / image sizes (REPROJ:77-137, getTransforms + getIntrinsics)
  explicit ReprojectionNode(std::shared_ptr<Context> ctx);
  // REPROJ:139-235 — one Person2DList per camera (topic <cam>/skel_pred): header.stamp = ts_per_cam[i], fb_delay = fb_delay_per_cam[i].
  // Returns false (nothing published) when frame_id != "base" or ts_per_cam has the wrong size (REPROJ:141-147).
  bool fusedSkeletonCallback(const person_msgs::PersonCovList& humans, std::vector<person_msgs::Person2DList>& humans_in_cam);

 private:
  std::shared_ptr<Context> ctx_;
};

}  // namespace se3d
