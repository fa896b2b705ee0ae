// se3d_nodes.cpp — see se3d_nodes.hpp.  Host plumbing only; no numerics of the path are evaluated here.
#include "se3d_nodes.hpp"
#include "se3d_exchange.hpp"

#include <chrono>
#include <cstring>
#include <iostream>

namespace se3d {

static const double g_max_sync_diff = 0.067;  // S3D:64
static const char* BASE_FRAME = "base";       // S3D:45

Context::Context(const std::vector<CameraCalib>& cams, const NodeParams& prm, int max_det_per_cam, int max_hyp, int device, const se3d_options* options, int max_frames) {
  if (options) opt_ = *options;
  max_frames_ = std::max(1, max_frames);
  n_cam_ = (int)cams.size();
  max_hyp_ = std::max(max_hyp, max_det_per_cam);   // the library wants max_hyp >= max_det_per_cam (the first camera seeds one hypothesis per detection)
  max_det_ = max_det_per_cam;
  device_ = device;
  prm_ = prm;
  cams_ = cams;
  if (n_cam_ < 2) throw std::runtime_error("Need at least 2 cameras for triangulation");  // S3D:1133-1136
  std::vector<double> T(16 * (size_t)n_cam_);
  Rt_.resize(12 * (size_t)n_cam_);
  K_.resize(4 * (size_t)n_cam_);
  F_.resize(9 * (size_t)n_cam_ * (n_cam_ - 1) / 2);
  for (int i = 0; i < n_cam_; ++i) {
    std::memcpy(&T[16 * (size_t)i], cams[i].T_base_to_cam.data(), sizeof(double) * 16);
    K_[4 * i + 0] = (float)cams[i].fx; K_[4 * i + 1] = (float)cams[i].fy; K_[4 * i + 2] = (float)cams[i].cx; K_[4 * i + 3] = (float)cams[i].cy;
  }
  if (se3d_rig_from_transforms(n_cam_, T.data(), Rt_.data(), F_.data()) != SE3D_OK) throw std::runtime_error("se3d_rig_from_transforms failed");
  ctx_ = create(max_det_, max_hyp_);
}

se3d_ctx* Context::create(int max_det, int max_hyp) const {
  se3d_rig rig{n_cam_, const_cast<float*>(Rt_.data()), const_cast<float*>(K_.data()), const_cast<float*>(F_.data())};
  se3d_params p{prm_.pose_method == "h36m" ? SE3D_POSE_H36M : SE3D_POSE_SIMPLE, prm_.norm_height ? 1 : 0, prm_.max_epi_dist};
  se3d_limits lim{max_frames_, max_det, max_hyp, max_hyp};
  if (se3d_abi_version() != SE3D_ABI_VERSION)   // the library's structs have another layout than the header this file was compiled with
    throw std::runtime_error("libse3d_hip.so has struct layout version " + std::to_string(se3d_abi_version()) + ", this host layer was built against " +
                             std::to_string(SE3D_ABI_VERSION) + " (include/se3d.h: SE3D_ABI_VERSION)");
  se3d_ctx* c = se3d_create_ex(&rig, &p, &lim, &opt_, device_);
  if (!c) throw std::runtime_error(std::string("se3d_create failed: ") + se3d_last_error(nullptr) + " (there is no CPU fallback)");
  return c;
}

void Context::grow(int min_det_per_cam, int min_hyp, const char* why) {
  if (min_det_per_cam > kCapacityMax || min_hyp > kCapacityMax)
    throw std::runtime_error(std::string(why) + ": needs " + std::to_string(std::max(min_det_per_cam, min_hyp)) +
                             " slots, the library's capacity is " + std::to_string(kCapacityMax) + " detections per camera / hypotheses / tracks");
  int det = max_det_, hyp = max_hyp_;
  while (det < min_det_per_cam) det = std::min(2 * det, kCapacityMax);
  while (hyp < std::max(min_hyp, det)) hyp = std::min(2 * hyp, kCapacityMax);
  if (det == max_det_ && hyp == max_hyp_) return;
  // the running pose_prior instance moves over: its tracks in vector order and the node globals
  std::vector<unsigned char> blob((size_t)se3d_refine_state_bytes(ctx_));
  int64_t n = 0;
  if (se3d_refine_save(ctx_, 0, blob.data(), (int64_t)blob.size(), &n) != SE3D_OK) throw std::runtime_error(std::string("se3d_refine_save: ") + se3d_last_error(ctx_));
  se3d_ctx* bigger = create(det, hyp);
  if (se3d_refine_load(bigger, 0, blob.data(), n) != SE3D_OK) {
    const std::string msg = std::string("se3d_refine_load: ") + se3d_last_error(bigger);
    se3d_destroy(bigger);
    throw std::runtime_error(msg);
  }
  if (!rp_T_.empty() && se3d_set_reprojection_cameras(bigger, rp_T_.data(), rp_P_.data(), rp_wh_.data()) != SE3D_OK) {
    const std::string msg = std::string("se3d_set_reprojection_cameras: ") + se3d_last_error(bigger);
    se3d_destroy(bigger);
    throw std::runtime_error(msg);
  }
  se3d_destroy(ctx_);
  ctx_ = bigger;
  std::cout << "se3d: " << why << ": limits raised to " << det << " detections per camera, " << hyp << " hypotheses / tracks (were " << max_det_ << ", "
            << max_hyp_ << ")" << std::endl;
  max_det_ = det;
  max_hyp_ = hyp;
  ++n_grown_;
}

void Context::set_reprojection_cameras(const std::vector<double>& T, const std::vector<double>& P, const std::vector<int32_t>& wh) {
  if (se3d_set_reprojection_cameras(ctx_, T.data(), P.data(), wh.data()) != SE3D_OK)
    throw std::runtime_error(std::string("se3d_set_reprojection_cameras: ") + se3d_last_error(ctx_));
  rp_T_ = T; rp_P_ = P; rp_wh_ = wh;
}
Context::~Context() { se3d_destroy(ctx_); }

static void to_pod(const person_msgs::Person2D& p, se3d_person2d& q) {
  std::memset(&q, 0, sizeof(q));
  q.score = p.score;
  for (int k = 0; k < 4; ++k) q.bbox[k] = p.bbox[k];
  const size_t n = p.keypoints.size() < SE3D_NUM_KEYPOINTS ? p.keypoints.size() : SE3D_NUM_KEYPOINTS;
  for (size_t k = 0; k < n; ++k) {
    const auto& kp = p.keypoints[k];
    q.keypoints[k].x = kp.x; q.keypoints[k].y = kp.y; q.keypoints[k].score = kp.score;
    for (int a = 0; a < 3; ++a) q.keypoints[k].cov[a] = kp.cov[a];
  }
}
static void from_pod(const se3d_person3d& s, person_msgs::PersonCov& d) {
  d.id = s.id;
  d.score = s.score;
  d.keypoints.resize(SE3D_NUM_FUSION_KEYPOINTS);  // S3D:686
  for (int k = 0; k < SE3D_NUM_FUSION_KEYPOINTS; ++k) {
    d.keypoints[k].joint.x = s.keypoints[k].joint[0]; d.keypoints[k].joint.y = s.keypoints[k].joint[1]; d.keypoints[k].joint.z = s.keypoints[k].joint[2];
    d.keypoints[k].score = s.keypoints[k].score;
    for (int a = 0; a < 6; ++a) d.keypoints[k].cov[a] = s.keypoints[k].cov[a];
  }
}
static void to_pod3(const person_msgs::PersonCov& s, se3d_person3d& d) {
  std::memset(&d, 0, sizeof(d));
  d.id = s.id;
  d.score = s.score;
  const size_t n = s.keypoints.size() < SE3D_NUM_FUSION_KEYPOINTS ? s.keypoints.size() : SE3D_NUM_FUSION_KEYPOINTS;
  for (size_t k = 0; k < n; ++k) {
    d.keypoints[k].joint[0] = s.keypoints[k].joint.x; d.keypoints[k].joint[1] = s.keypoints[k].joint.y; d.keypoints[k].joint[2] = s.keypoints[k].joint.z;
    d.keypoints[k].score = s.keypoints[k].score;
    for (int a = 0; a < 6; ++a) d.keypoints[k].cov[a] = s.keypoints[k].cov[a];
  }
}

void flatten_frame(const std::vector<person_msgs::Person2DList>& people, double stamp, FlatFrame& out) {
  const int N = (int)people.size();
  out.off.assign(N + 1, 0);
  out.flat.clear();
  out.fb.assign(N, -1.f);
  for (int i = 0; i < N; ++i) {
    for (const auto& p : people[i].persons) { out.flat.emplace_back(); to_pod(p, out.flat.back()); }
    out.off[i + 1] = (int32_t)out.flat.size();
    out.fb[i] = people[i].fb_delay;
  }
  out.in = se3d_frame_in{N, out.off.data(), out.flat.empty() ? nullptr : out.flat.data(), out.fb.data(), stamp};
}

void Skeleton3DNode::triangulate_persons(const std::vector<person_msgs::Person2DList>& people, person_msgs::PersonCovList& persons3d_msg,
                                         std::vector<int32_t>* assoc, int* n_hyp) {
  const int N = ctx_->n_cam();
  if ((int)people.size() != N) throw std::runtime_error("people.size() != NUM_CAMERAS");  // assert at S3D:534
  std::vector<int32_t> off(N + 1, 0);
  std::vector<se3d_person2d> flat;
  std::vector<float> fb(N, -1.f);
  for (int i = 0; i < N; ++i) {
    for (const auto& p : people[i].persons) { flat.emplace_back(); to_pod(p, flat.back()); }
    off[i + 1] = (int32_t)flat.size();
    fb[i] = people[i].fb_delay;
  }
  int max_per_cam = 0;
  for (int i = 0; i < N; ++i) max_per_cam = std::max(max_per_cam, off[i + 1] - off[i]);
  if (max_per_cam > ctx_->max_det_per_cam()) {
    if (shard_) throw std::runtime_error("hypothesis shard: more detections in one camera than limits.max_det_per_cam (the limits are fixed in this mode)");
    ctx_->grow(max_per_cam, 0, "more detections in one camera than limits.max_det_per_cam");
  }
  std::vector<se3d_person3d> out;
  std::vector<int32_t> hyp, asc;
  se3d_frame_in in{N, off.data(), flat.empty() ? nullptr : flat.data(), fb.data(), persons3d_msg.header.stamp};
  se3d_frame_out res{};
  for (;;) {
    const int H = ctx_->max_hyp();
    out.assign(H, se3d_person3d{});
    hyp.assign(H, 0);
    asc.assign((size_t)H * N, -1);
    res = se3d_frame_out{H, H, 0, 0, 0u, out.data(), hyp.data(), asc.data()};
    if (shard_) {   // hypotheses split over the ranks: stage the frame, association + own triangulation, all-gather, post-processing, fetch
      if (se3d_stage_frames(ctx_->get(), 1, &in) != SE3D_OK) throw std::runtime_error(std::string("se3d_stage_frames: ") + se3d_last_error(ctx_->get()));
      shard_->run_stage1();
      if (se3d_fetch_stage1(ctx_->get(), 1, &res) != SE3D_OK) throw std::runtime_error(std::string("se3d_fetch_stage1: ") + se3d_last_error(ctx_->get()));
      if (res.flags & SE3D_FLAG_HYP_OVERFLOW) throw std::runtime_error("hypothesis shard: SE3D_FLAG_HYP_OVERFLOW (the limits are fixed in this mode: start with a larger max_hyp)");
      break;
    }
    const int rc = se3d_triangulate(ctx_->get(), &in, &res);
    if (rc != SE3D_OK) throw std::runtime_error(std::string("se3d_triangulate: ") + se3d_last_error(ctx_->get()));
    if (!(res.flags & SE3D_FLAG_HYP_OVERFLOW)) break;
    // the sweep spawned more hypotheses than limits.max_hyp and truncated (the reference appends, S3D:636-673): raise the limit, run the frame again
    ctx_->grow(0, 2 * H, "SE3D_FLAG_HYP_OVERFLOW (more hypotheses than limits.max_hyp)");
  }
  if (res.flags & SE3D_FLAG_ASSIGNMENT_FAILED) throw std::runtime_error("se3d_triangulate: SE3D_FLAG_ASSIGNMENT_FAILED (non-finite association costs)");
  persons3d_msg.persons.clear();  // < 2 cameras with detections: the empty list is still published (S3D:557-560, 1071)
  for (int p = 0; p < res.n_persons; ++p) { persons3d_msg.persons.emplace_back(); from_pod(out[p], persons3d_msg.persons.back()); }
  if (assoc) assoc->assign(asc.begin(), asc.begin() + (size_t)res.n_hyp * N);
  if (n_hyp) *n_hyp = res.n_hyp;
}

bool Skeleton3DNode::skeletonThreadStep(std::vector<person_msgs::Person2DList> people, person_msgs::PersonCovList& persons3d_msg) {
  const int N = ctx_->n_cam();
  if ((int)people.size() != N) throw std::runtime_error("people.size() != NUM_CAMERAS");  // assert at S3D:1027
  double t_max = 0.0;  // most recent message (pivot element), S3D:1029-1036
  int t_max_idx = -1;
  for (int i = 0; i < N; ++i)
    if (people[i].header.stamp > t_max) { t_max = people[i].header.stamp; t_max_idx = i; }
  if (t_max_idx < 0) return false;
  const auto t1 = std::chrono::high_resolution_clock::now();
  const double delta_t = people[t_max_idx].header.stamp - last_stamp_;
  if (delta_t > 0.17) std::cout << "WARNING: Large frame delay delta_t = " << delta_t << "s (should be < 0.17s)" << std::endl;  // S3D:1040-1042
  if (delta_t <= 0.0) {  // S3D:1043-1046
    std::cout << "WARNING: re-using message or jumped backwards in time: delta_t = " << delta_t << "s. Will not process this message!" << std::endl;
    return false;
  }
  last_stamp_ = people[t_max_idx].header.stamp;
  for (int i = 0; i < N; ++i) {  // S3D:1049-1057: a view older than 67 ms w.r.t. the pivot is replaced by an empty dummy
    const double dt = t_max - people[i].header.stamp;
    if (dt > g_max_sync_diff) {
      person_msgs::Person2DList dummy;
      dummy.header = people[i].header;
      dummy.fb_delay = people[i].fb_delay;
      people[i] = dummy;
    }
  }
  persons3d_msg = person_msgs::PersonCovList();
  persons3d_msg.header = people[t_max_idx].header;  // S3D:1060-1065
  for (int i = 0; i < N; ++i) {
    persons3d_msg.ts_per_cam.push_back(people[i].header.stamp);
    persons3d_msg.ts_per_cam_ns.push_back(people[i].header.stamp_ns);
    persons3d_msg.fb_delay_per_cam.push_back(people[i].fb_delay);
  }
  persons3d_msg.header.frame_id = BASE_FRAME;
  triangulate_persons(people, persons3d_msg);
  const auto t2 = std::chrono::high_resolution_clock::now();
  const double ms = std::chrono::duration_cast<std::chrono::microseconds>(t2 - t1).count() / 1000.;  // S3D:1076-1084
  timings_[0] += ms;
  ++timing_cnt_[0];
  const size_t n = persons3d_msg.persons.size();
  if (n > 0 && n < timings_.size()) { timings_[n] += ms; ++timing_cnt_[n]; }
  return true;
}

void PosePriorNode::reset() { se3d_refine_reset(ctx_->get()); n_tracks_ = 0; }

void PosePriorNode::skeletonCallback(const person_msgs::PersonCovList& persons, person_msgs::PersonCovList& persons_fused,
                                     person_msgs::PersonCovList& persons_fused_pred) {
  std::vector<se3d_person3d> dets(persons.persons.size());
  for (size_t i = 0; i < dets.size(); ++i) to_pod3(persons.persons[i], dets[i]);
  // g_tracks.push_back (PP:570-580) never fails in the reference.  A callback adds at most one track per detection, so the live tracks
  // after it are at most n_tracks_ + n_det: raise the limits BEFORE the call when that bound does not fit (a truncated step could not
  // be undone: it has already changed the track state).
  const int need = std::max((int)dets.size(), n_tracks_ + (int)dets.size());
  if (need > ctx_->max_hyp()) {
    if (shard_) throw std::runtime_error("hypothesis shard: more detections + live tracks than limits.max_hyp / max_tracks (the limits are fixed in this mode)");
    ctx_->grow(0, need, "more detections + live tracks than limits.max_hyp / max_tracks");
  }
  const int H = ctx_->max_hyp();
  std::vector<se3d_person3d> fused(H), pred(H);
  std::vector<int32_t> iters(H);
  se3d_fused_out fo{H, 0, 0, 0.f, fused.data(), pred.data(), iters.data(), 0u, nullptr};
  int rc;
  if (shard_) {   // detections split over the ranks: the resident stage-1 persons of frame slot 0 (the list in `persons`), fb_delay as staged with the frame
    const double t = persons.header.stamp;
    shard_->run_stage2(&t, 0);
    rc = se3d_fetch_stage2(ctx_->get(), 1, &fo);
  } else {
    rc = se3d_refine(ctx_->get(), seq_, (int32_t)dets.size(), dets.empty() ? nullptr : dets.data(), (int32_t)persons.fb_delay_per_cam.size(),
                     persons.fb_delay_per_cam.empty() ? nullptr : persons.fb_delay_per_cam.data(), persons.header.stamp, &fo);
  }
  if (rc != SE3D_OK) throw std::runtime_error(std::string("se3d_refine: ") + se3d_last_error(ctx_->get()));
  if (fo.flags & SE3D_S2FLAG_TRACK_OVERFLOW) throw std::runtime_error("se3d_refine: SE3D_S2FLAG_TRACK_OVERFLOW although the limits were raised beforehand");
  if (fo.flags & SE3D_S2FLAG_ASSIGNMENT_FAILED) throw std::runtime_error("se3d_refine: SE3D_S2FLAG_ASSIGNMENT_FAILED (non-finite track association costs)");
  n_tracks_ = fo.n_tracks;
  persons_fused = person_msgs::PersonCovList();
  persons_fused.header = persons.header;  // PP:528-532
  persons_fused.ts_per_cam = persons.ts_per_cam;
  persons_fused.ts_per_cam_ns = persons.ts_per_cam_ns;
  persons_fused.fb_delay_per_cam.assign(persons.fb_delay_per_cam.size(), fo.pred_delta_t);
  persons_fused_pred = persons_fused;
  for (int p = 0; p < fo.n_persons; ++p) {
    persons_fused.persons.emplace_back(); from_pod(fused[p], persons_fused.persons.back());
    persons_fused_pred.persons.emplace_back(); from_pod(pred[p], persons_fused_pred.persons.back());
  }
}

ReprojectionNode::ReprojectionNode(std::shared_ptr<Context> ctx) : ctx_(std::move(ctx)) {
  const auto& cams = ctx_->cameras();
  std::vector<double> T(16 * cams.size()), P(4 * cams.size());
  std::vector<int32_t> wh(2 * cams.size());
  for (size_t i = 0; i < cams.size(); ++i) {
    std::memcpy(&T[16 * i], cams[i].T_base_to_cam.data(), sizeof(double) * 16);
    P[4 * i] = cams[i].fx; P[4 * i + 1] = cams[i].fy; P[4 * i + 2] = cams[i].cx; P[4 * i + 3] = cams[i].cy;
    wh[2 * i] = cams[i].width; wh[2 * i + 1] = cams[i].height;
  }
  ctx_->set_reprojection_cameras(T, P, wh);   // kept by the context: re-applied when its limits are raised
}

bool ReprojectionNode::fusedSkeletonCallback(const person_msgs::PersonCovList& humans, std::vector<person_msgs::Person2DList>& humans_in_cam) {
  const int n_cam = ctx_->n_cam();
  if (humans.header.frame_id != "base") return false;                       // REPROJ:141-143
  if ((int)humans.ts_per_cam.size() != n_cam || (int)humans.fb_delay_per_cam.size() != n_cam) return false;  // REPROJ:147
  std::vector<se3d_person3d> in;
  in.reserve(humans.persons.size());
  for (const auto& p : humans.persons) {
    if (p.keypoints.size() != SE3D_NUM_FUSION_KEYPOINTS) continue;          // REPROJ:151
    in.emplace_back();
    to_pod3(p, in.back());
  }
  std::vector<int32_t> off(n_cam + 1, 0);
  std::vector<se3d_person2d> flat((size_t)n_cam * (in.empty() ? 1 : in.size()));
  se3d_reproj_out out{(int32_t)flat.size(), off.data(), flat.data(), nullptr};
  const int rc = se3d_reproject(ctx_->get(), (int32_t)in.size(), in.empty() ? nullptr : in.data(), &out);
  if (rc != SE3D_OK) throw std::runtime_error(std::string("se3d_reproject: ") + se3d_last_error(ctx_->get()));
  humans_in_cam.assign(n_cam, person_msgs::Person2DList());
  for (int i = 0; i < n_cam; ++i) {
    auto& l = humans_in_cam[i];
    l.header.stamp = humans.ts_per_cam[i];                                   // REPROJ:230-231
    l.header.stamp_ns = (int)humans.ts_per_cam_ns.size() == n_cam ? humans.ts_per_cam_ns[i] : -1;
    l.header.frame_id = "cam_" + std::to_string(i);
    l.fb_delay = humans.fb_delay_per_cam[i];
    for (int k = off[i]; k < off[i + 1]; ++k) {
      person_msgs::Person2D p;
      p.score = flat[k].score;
      for (int a = 0; a < 4; ++a) p.bbox[a] = flat[k].bbox[a];
      p.keypoints.resize(SE3D_NUM_KEYPOINTS);
      for (int j = 0; j < SE3D_NUM_KEYPOINTS; ++j) {
        const auto& s = flat[k].keypoints[j];
        p.keypoints[j].x = s.x; p.keypoints[j].y = s.y; p.keypoints[j].score = s.score;
        for (int a = 0; a < 3; ++a) p.keypoints[j].cov[a] = s.cov[a];
      }
      l.persons.push_back(std::move(p));
    }
  }
  return true;
}

}  // namespace se3d
