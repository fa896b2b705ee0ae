// se3d_replay.cpp — ROS-free replay harness: feeds recorded/synthetic per-camera Person2DList streams through the C++ host
// layer (Skeleton3DNode::skeletonThreadStep -> PosePriorNode::skeletonCallback) exactly like the node pipeline of
// pose_triangulate_demo.launch would, and dumps what the nodes would publish.  Used by tests/test_gpu_host_cpp.py.
//   se3d_replay <scene.bin> <out.bin> [reproj.bin [width height]] [--ranks N [--backend rccl|staged]]
// With a third path the predicted skeletons are also fed through ReprojectionNode::fusedSkeletonCallback and the per-camera
// Person2DList payloads (<cam>/skel_pred) are dumped.
// --ranks N (round 6; north_star config 5 in the C++ host layer): the process starts N fresh children — one per GPU, before anything has touched
// the GPU — that all replay the scene with the hypotheses / detections split over the ranks (se3d::HypothesisShard: RCCL all-gathers on the
// contexts' streams; --backend staged: the test-only host-staged collective, all ranks on GPU 0).  Rank r writes <out.bin>.rank<r>; every rank's
// file must equal the single-process <out.bin> byte for byte (tests/test_gpu_host_cpp.py).
// --walk-exchange (with --ranks): the throughput sharding's exchange instead — every rank stages the scene's frames as ONE resident batch, runs stage 1 and the
// stage-2 walk over them, packs the published persons densely on the device and all-gathers them (se3d::SkeletonExchange); every rank checks that its own block
// equals what se3d_fetch_stage2_walk returns slot by slot and that every other rank's block equals its own (all ranks replay the same scene).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>

#include <sys/wait.h>
#include <unistd.h>

#include "se3d_exchange.hpp"
#include "se3d_markers.hpp"
#include "se3d_nodes.hpp"

template <typename T> static T rd(std::ifstream& f) { T v; f.read(reinterpret_cast<char*>(&v), sizeof(T)); return v; }
template <typename T> static void wr(std::ofstream& f, const T& v) { f.write(reinterpret_cast<const char*>(&v), sizeof(T)); }

static void write_list(std::ofstream& o, const person_msgs::PersonCovList& l) {
  wr<double>(o, l.header.stamp);
  wr<int32_t>(o, (int32_t)l.fb_delay_per_cam.size());
  for (float v : l.fb_delay_per_cam) wr<float>(o, v);
  for (double v : l.ts_per_cam) wr<double>(o, v);
  wr<int32_t>(o, (int32_t)l.persons.size());
  for (const auto& p : l.persons) {
    wr<uint32_t>(o, p.id);
    for (const auto& k : p.keypoints) {
      wr<double>(o, k.joint.x); wr<double>(o, k.joint.y); wr<double>(o, k.joint.z);
      wr<float>(o, k.score);
      for (double c : k.cov) wr<double>(o, c);
    }
  }
}

// --ranks N: start the ranks as fresh child processes (fork + exec of this binary with SE3D_RANK / SE3D_WORLD / SE3D_RDV_DIR in the environment) and
// wait for them.  Nothing in this process has touched the GPU.
static int launch_ranks(int n, const std::string& backend, char** argv) {
  char tmpl[] = "/tmp/se3d_replay_rdv_XXXXXX";
  const char* dir = mkdtemp(tmpl);
  if (!dir) { std::perror("mkdtemp"); return 2; }
  std::vector<pid_t> pids;
  for (int r = 0; r < n; ++r) {
    const pid_t pid = fork();
    if (pid < 0) { std::perror("fork"); return 2; }
    if (pid == 0) {
      setenv("SE3D_RANK", std::to_string(r).c_str(), 1);
      setenv("SE3D_WORLD", std::to_string(n).c_str(), 1);
      setenv("SE3D_RDV_DIR", dir, 1);
      setenv("SE3D_BACKEND", backend.c_str(), 1);
      setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
      execv(argv[0], argv);
      std::perror("execv");
      _exit(127);
    }
    pids.push_back(pid);
  }
  int worst = 0;
  for (pid_t pid : pids) {
    int st = 0;
    (void)waitpid(pid, &st, 0);
    const int code = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
    worst = std::max(worst, code);
  }
  const std::string rm = std::string("rm -rf '") + dir + "'";
  if (std::system(rm.c_str()) != 0) std::fprintf(stderr, "could not remove %s\n", dir);
  return worst;
}

int main(int argc_all, char** argv_all) {
  // options anywhere on the line; the positional arguments keep their places
  std::vector<char*> pos;
  int ranks = 0;
  bool walk_exchange = false;
  std::string backend = "rccl";
  for (int i = 0; i < argc_all; ++i) {
    if (!std::strcmp(argv_all[i], "--ranks") && i + 1 < argc_all) ranks = std::atoi(argv_all[++i]);
    else if (!std::strcmp(argv_all[i], "--backend") && i + 1 < argc_all) backend = argv_all[++i];
    else if (!std::strcmp(argv_all[i], "--walk-exchange")) walk_exchange = true;
    else pos.push_back(argv_all[i]);
  }
  const int argc = (int)pos.size();
  char** argv = pos.data();
  if (argc < 3) { std::fprintf(stderr, "usage: %s scene.bin out.bin [reproj.bin [width height]] [--ranks N [--backend rccl|staged]]\n", argv[0]); return 2; }
  const bool is_rank = std::getenv("SE3D_RANK") != nullptr;
  if (ranks >= 1 && !is_rank) return launch_ranks(ranks, backend, argv_all);
  const int my_rank = is_rank ? std::atoi(std::getenv("SE3D_RANK")) : 0, world = is_rank ? std::atoi(std::getenv("SE3D_WORLD")) : 1;
  if (is_rank) backend = std::getenv("SE3D_BACKEND") ? std::getenv("SE3D_BACKEND") : backend;
  std::ifstream in(argv[1], std::ios::binary);
  char magic[8];
  in.read(magic, 8);
  if (std::memcmp(magic, "SE3DSCN1", 8) != 0) { std::fprintf(stderr, "bad scene file\n"); return 2; }
  const int n_cam = rd<int32_t>(in);
  std::vector<se3d::CameraCalib> cams(n_cam);
  for (auto& c : cams) {
    for (double& v : c.T_base_to_cam) v = rd<double>(in);
    c.fx = rd<double>(in); c.fy = rd<double>(in); c.cx = rd<double>(in); c.cy = rd<double>(in);
  }
  if (argc >= 6)
    for (auto& c : cams) { c.width = std::atoi(argv[4]); c.height = std::atoi(argv[5]); }
  se3d::NodeParams prm;
  prm.pose_method = rd<int32_t>(in) == 1 ? "h36m" : "simple";
  prm.norm_height = rd<int32_t>(in) != 0;
  prm.max_epi_dist = rd<double>(in);
  const int n_frames = rd<int32_t>(in);
  try {
    // initial limits (SE3D_REPLAY_MAX_DET / SE3D_REPLAY_MAX_HYP: tests start far below what the scene needs; the nodes raise them)
    const int det0 = std::getenv("SE3D_REPLAY_MAX_DET") ? std::atoi(std::getenv("SE3D_REPLAY_MAX_DET")) : 16;
    const int hyp0 = std::getenv("SE3D_REPLAY_MAX_HYP") ? std::atoi(std::getenv("SE3D_REPLAY_MAX_HYP")) : 64;
    const int device = is_rank && backend != "staged" ? my_rank : 0;   // one GPU per rank; the test-only staged backend shares GPU 0
    // SE3D_REPLAY_REFINE_STEPS=1: se3d_refine as the five per-step launches instead of one fused-walk launch (se3d_options.refine_steps) — the kernels the
    // hypothesis-shard calls use, so that a sharded replay can be compared with a single-process one byte for byte (the two LM forms sum in another order)
    se3d_options opt{};
    opt.refine_steps = std::getenv("SE3D_REPLAY_REFINE_STEPS") ? std::atoi(std::getenv("SE3D_REPLAY_REFINE_STEPS")) : 0;
    auto ctx = std::make_shared<se3d::Context>(cams, prm, det0, hyp0, device, &opt);
    se3d::Skeleton3DNode skel(ctx);
    se3d::PosePriorNode prior(ctx);
    std::unique_ptr<se3d::Collective> coll;
    std::unique_ptr<se3d::HypothesisShard> shard;
    if (is_rank) {
      const std::string dir = std::getenv("SE3D_RDV_DIR") ? std::getenv("SE3D_RDV_DIR") : "/tmp";
      coll = backend == "staged" ? se3d::make_staged_collective(my_rank, world, dir) : se3d::make_rccl_collective(my_rank, world, device, dir);
      shard.reset(new se3d::HypothesisShard(ctx->get(), *coll, 1, 1));   // one frame and one sequence per callback: the reference's operating point
      skel.set_shard(shard.get());
      prior.set_shard(shard.get());
      prior.reset();
      std::cout << "rank " << my_rank << " of " << world << " on device " << device << ", collective: " << coll->name() << ", all-gather bytes per frame: "
                << shard->allgather_bytes_stage1() << " + " << shard->allgather_bytes_stage2() << std::endl;
    }
    if (walk_exchange) {
      if (!is_rank) { std::fprintf(stderr, "--walk-exchange needs --ranks N\n"); return 2; }
      if (n_frames > 64) { std::fprintf(stderr, "--walk-exchange: at most 64 frames (one walk)\n"); return 2; }
      shard.reset();   // (the constructor above switched the hypothesis shard on for ctx; this mode uses a context of its own)
      skel.set_shard(nullptr); prior.set_shard(nullptr);
      auto wctx = std::make_shared<se3d::Context>(cams, prm, det0, hyp0, device, &opt, n_frames);
      std::vector<se3d::FlatFrame> ff(n_frames);
      std::vector<se3d_frame_in> fin(n_frames);
      std::vector<double> ts(n_frames);
      std::vector<int32_t> slots(n_frames);
      for (int f = 0; f < n_frames; ++f) {
        std::vector<person_msgs::Person2DList> people(n_cam);
        double stamp = 0.0;
        for (int c = 0; c < n_cam; ++c) {
          people[c].header.stamp = rd<double>(in);
          stamp = std::max(stamp, people[c].header.stamp);
          people[c].fb_delay = rd<float>(in);
          const int np = rd<int32_t>(in);
          people[c].persons.resize(np);
          for (auto& p : people[c].persons) {
            p.score = rd<float>(in);
            for (float& b : p.bbox) b = rd<float>(in);
            p.keypoints.resize(17);
            for (auto& k : p.keypoints) { k.x = rd<float>(in); k.y = rd<float>(in); k.score = rd<float>(in); for (float& cv : k.cov) cv = rd<float>(in); }
          }
        }
        se3d::flatten_frame(people, stamp, ff[f]);
        fin[f] = ff[f].in;
        ts[f] = stamp;
        slots[f] = f;
      }
      se3d_ctx* c = wctx->get();
      auto chk = [&](int rc, const char* what) { if (rc != SE3D_OK) throw std::runtime_error(std::string(what) + ": " + se3d_last_error(c)); };
      chk(se3d_stage_frames(c, n_frames, fin.data()), "se3d_stage_frames");
      chk(se3d_run_stage1(c), "se3d_run_stage1");
      chk(se3d_refine_reset(c), "se3d_refine_reset");
      chk(se3d_run_stage2_walk(c, 1, n_frames, slots.data(), ts.data()), "se3d_run_stage2_walk");
      const int cap = n_frames * hyp0;
      se3d::SkeletonExchange ex(c, *coll, n_frames, cap);
      ex.exchange();
      std::vector<std::vector<se3d_person3d>> persons;
      std::vector<std::vector<int32_t>> offsets;
      ex.fetch(persons, offsets);
      // own block against the host fetch, slot by slot
      size_t total = 0;
      std::vector<se3d_person3d> fused(hyp0), pred(hyp0);
      std::vector<int32_t> iters(hyp0);
      for (int f = 0; f < n_frames; ++f) {
        se3d_fused_out fo{hyp0, 0, 0, 0.f, fused.data(), pred.data(), iters.data(), 0u, nullptr};
        chk(se3d_fetch_stage2_walk(c, f, 1, &fo), "se3d_fetch_stage2_walk");
        const int o0 = offsets[my_rank][f], o1 = offsets[my_rank][f + 1];
        if (o1 - o0 != fo.n_persons || (fo.n_persons && std::memcmp(&persons[my_rank][o0], fused.data(), sizeof(se3d_person3d) * (size_t)fo.n_persons) != 0))
          throw std::runtime_error("walk exchange: frame slot " + std::to_string(f) + " of the own block differs from se3d_fetch_stage2_walk");
        total += (size_t)fo.n_persons;
      }
      if (offsets[my_rank][n_frames] != (int32_t)total || offsets[my_rank][n_frames + 1] != 0) throw std::runtime_error("walk exchange: totals of the own block are wrong");
      for (int r = 0; r < world; ++r)
        if (offsets[r] != offsets[my_rank] || persons[r].size() != persons[my_rank].size() ||
            (total && std::memcmp(persons[r].data(), persons[my_rank].data(), sizeof(se3d_person3d) * total) != 0))
          throw std::runtime_error("walk exchange: the block of rank " + std::to_string(r) + " differs from this rank's (all ranks replay the same scene)");
      std::cout << "walk exchange ok: rank " << my_rank << " of " << world << ", " << total << " published persons in " << n_frames << " frame slots from every rank, collective: "
                << coll->name() << std::endl;
      return 0;
    }
    std::ofstream out(is_rank ? std::string(argv[2]) + ".rank" + std::to_string(my_rank) : std::string(argv[2]), std::ios::binary);
    wr<int32_t>(out, n_frames);
    std::unique_ptr<se3d::ReprojectionNode> reproj;
    std::ofstream rout;
    if (argc >= 4) {
      reproj.reset(new se3d::ReprojectionNode(ctx));
      rout.open(argv[3], std::ios::binary);
      wr<int32_t>(rout, n_frames);
      wr<int32_t>(rout, n_cam);
    }
    int n_tuples = 0;
    se3d::FusedMarkerState marker_state;   // the nodes' MarkerArray publications (S3D:1172-1174, PP:945): built per frame, totals reported below
    size_t n_markers3d = 0, n_markers_fused = 0, n_markers_deleted = 0;
    const bool vis_cov = std::getenv("SE3D_REPLAY_VIS_COV") != nullptr;
    se3d::SigmaRange sigma_range;
    auto handle_tuple = [&](const std::vector<person_msgs::Person2DList>& people) {
      ++n_tuples;
      person_msgs::PersonCovList persons3d, fused, pred;
      const bool processed = skel.skeletonThreadStep(people, persons3d);
      wr<int32_t>(out, processed ? 1 : 0);
      if (!processed) return;
      prior.skeletonCallback(persons3d, fused, pred);
      write_list(out, persons3d);
      write_list(out, fused);
      write_list(out, pred);
      {
        visualization_msgs::MarkerArray skel3d_msg, vis_markers;
        se3d::skeleton_markers(persons3d, prm.pose_method, skel3d_msg);
        se3d::fused_markers(fused, vis_markers);
        if (vis_cov) {   // ROS parameter `vis_cov` of both nodes (S3D:1096, PP:932): covariance ellipsoids next to the skeletons
          se3d::covariance_markers_3d(persons3d, prm.pose_method, skel3d_msg, &sigma_range);
          se3d::covariance_markers_fused(fused, vis_markers);
        }
        const size_t n_add = vis_markers.markers.size();
        marker_state.update(fused, vis_markers);
        n_markers3d += skel3d_msg.markers.size(); n_markers_fused += n_add; n_markers_deleted += vis_markers.markers.size() - n_add;
      }
      if (reproj) {
        pred.header.frame_id = "base";
        std::vector<person_msgs::Person2DList> per_cam;
        const bool ok = reproj->fusedSkeletonCallback(pred, per_cam);
        wr<int32_t>(rout, ok ? 1 : 0);
        if (ok)
          for (const auto& l : per_cam) {
            wr<double>(rout, l.header.stamp); wr<float>(rout, l.fb_delay); wr<int32_t>(rout, (int32_t)l.persons.size());
            for (const auto& p : l.persons) {
              wr<float>(rout, p.score);
              for (float b : p.bbox) wr<float>(rout, b);
              for (const auto& k : p.keypoints) { wr<float>(rout, k.x); wr<float>(rout, k.y); wr<float>(rout, k.score); for (float cv : k.cov) wr<float>(rout, cv); }
            }
          }
      }
    };
    std::unique_ptr<se3d::Skeleton3DFrontEnd> front;
    if (std::getenv("SE3D_REPLAY_SYNC"))
      front.reset(new se3d::Skeleton3DFrontEnd(n_cam, [&](const se3d::Skeleton3DFrontEnd::Tuple& t) {
        std::vector<person_msgs::Person2DList> people;
        for (const auto& m : t) people.push_back(*m);
        handle_tuple(people);
      }));
    for (int f = 0; f < n_frames; ++f) {
      std::vector<person_msgs::Person2DList> people(n_cam);
      for (int c = 0; c < n_cam; ++c) {
        people[c].header.stamp = rd<double>(in);
        people[c].fb_delay = rd<float>(in);
        const int np = rd<int32_t>(in);
        people[c].persons.resize(np);
        for (auto& p : people[c].persons) {
          p.score = rd<float>(in);
          for (float& b : p.bbox) b = rd<float>(in);
          p.keypoints.resize(17);
          for (auto& k : p.keypoints) { k.x = rd<float>(in); k.y = rd<float>(in); k.score = rd<float>(in); for (float& cv : k.cov) cv = rd<float>(in); }
        }
      }
      if (front) {  // SE3D_REPLAY_SYNC=1: the messages of this frame arrive one by one at the approximate-time synchroniser
        for (int c = 0; c < n_cam; ++c) front->add(std::make_shared<const person_msgs::Person2DList>(people[c]), c);
      } else {
        handle_tuple(people);
      }
    }
    out.seekp(0);
    wr<int32_t>(out, n_tuples);  // tuples actually handled (with the synchroniser in front the last frames may still be held back)
    std::cout << "Limits: raised " << ctx->n_grown() << " times, now " << ctx->max_det_per_cam() << " detections per camera, " << ctx->max_hyp() << " hypotheses / tracks" << std::endl;
    std::cout << "Markers: " << n_markers3d << " skeleton_3d, " << n_markers_fused << " fused, " << n_markers_deleted << " deleted" << std::endl;
    if (vis_cov && sigma_range.max_ev[0] > 0) {   // S3D:1243-1247
      std::cout << "Sigmas 3D [mm]:" << std::endl;
      std::cout << "min: x: " << sigma_range.min_mm(0) << "mm, y: " << sigma_range.min_mm(1) << "mm, z: " << sigma_range.min_mm(2) << "mm." << std::endl;
      std::cout << "max: x: " << sigma_range.max_mm(0) << "mm, y: " << sigma_range.max_mm(1) << "mm, z: " << sigma_range.max_mm(2) << "mm." << std::endl;
    }
    for (size_t i = 0; i < skel.timings_ms().size(); ++i)  // S3D:1234-1241
      if (skel.timing_counts()[i] > 0) {
        std::cout << "Triangulation: ";
        if (i > 0) std::cout << i << " detections: ";
        std::cout << "avg runtime: " << skel.timings_ms()[i] / skel.timing_counts()[i] << "ms" << std::endl;
      }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "se3d_replay: %s\n", e.what());
    return 1;
  }
  return 0;
}
