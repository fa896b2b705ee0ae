// skeleton_3d_node.cpp — catkin wrapper of the accelerated skeleton_3d node: the ROS surface of the reference's main()
// (skeleton_3d/src/skeleton_3d_triang_mult_node.cpp:1088-1248) around se3d::Skeleton3DNode (host/se3d_nodes.hpp -> libse3d_hip.so).
//   in : <cam>/human_joints (person_msgs/Person2DList, queue 1, tcpNoDelay; S3D:48, 1176-1179), <cam>/color/camera_info once, tf base -> <cam>_color_optical_frame
//   out: human_pose_estimation/persons_3d (person_msgs/PersonCovList, queue 1), human_pose_estimation/skeleton_3d (MarkerArray, queue 1)   S3D:53-54, 1172-1174
//   private parameters: pose_method ("simple"), vis_cov (false), max_epi_dist (0.050), cameras ([]: cam_1 .. cam_4)                          S3D:1095-1130
// Threads as in the reference: the spinner feeds the approximate-time synchroniser (queue max(3, 1 + N/4), 20 ms lower bound, age penalty 2:
// S3D:1218-1222; here the ROS-free restatement host/se3d_sync.hpp) whose tuples go into a latest-wins single-slot mailbox (S3D:999-1006);
// one worker thread runs skeletonThreadStep (pivot, frame drop, stale-view replacement, bookkeeping: S3D:1029-1084) and publishes.
// Built only inside a catkin workspace (CMakeLists.txt of this directory); never compiled in the ROS-free image.
#include <condition_variable>
#include <mutex>
#include <thread>

#include "ros_bridge.hpp"

namespace {
std::mutex g_mutex;
std::condition_variable g_cv;
std::vector<se3d_person_msgs::Person2DList> g_people;
bool g_updated = false;
}  // namespace

int main(int argc, char** argv) {
  ros::init(argc, argv, "skeleton_singlePerson_3d");   // node name of the reference (S3D:1091)
  ros::NodeHandle nh;
  ros::NodeHandle nh_private("~");

  se3d::NodeParams prm;
  bool vis_cov = false;
  nh_private.param<std::string>("pose_method", prm.pose_method, "simple");
  nh_private.param<bool>("vis_cov", vis_cov, false);
  nh_private.param<double>("max_epi_dist", prm.max_epi_dist, 0.050);
  int device = 0, max_det = 16, max_hyp = 64;
  nh_private.param<int>("device", device, 0);                    // (new) HIP device of this node
  nh_private.param<int>("max_det_per_cam", max_det, 16);         // (new) initial limits; the context grows when a frame exceeds them
  nh_private.param<int>("max_hyp", max_hyp, 64);
  const std::vector<std::string> cams = se3d_ros::camera_names(nh_private);
  const int N = (int)cams.size();
  ROS_INFO("NUM_CAMERAS: %d", N);
  if (N < 2) {
    ROS_ERROR("Need at least 2 cameras for triangulation. Aborting!");   // S3D:1133-1136
    return -1;
  }

  ros::Publisher pub_pers3d = nh.advertise<person_msgs::PersonCovList>("human_pose_estimation/persons_3d", 1);
  ros::Publisher pub_skel3d = nh.advertise<visualization_msgs::MarkerArray>("human_pose_estimation/skeleton_3d", 1);

  tf2_ros::Buffer tfBuffer;
  tf2_ros::TransformListener tfListener(tfBuffer);
  const auto T = se3d_ros::get_transforms(cams, tfBuffer);
  const auto intr = se3d_ros::get_intrinsics(cams, nh);
  std::shared_ptr<se3d::Context> ctx;
  try {
    ctx = std::make_shared<se3d::Context>(se3d_ros::calibration(T, intr), prm, max_det, max_hyp, device);   // rig tensors (S3D:1187-1211) + se3d_create
  } catch (const std::exception& e) {
    ROS_FATAL("%s", e.what());   // no GPU / no memory: there is no CPU fallback
    return -1;
  }
  ROS_INFO("Synchronizing %d cameras, calculated %zu fundamental matrices", N, ctx->fundamental().size() / 9);
  se3d::Skeleton3DNode node(ctx);
  se3d::SigmaRange sigma_range;

  std::thread worker([&] {   // skeletonThreadCallback (S3D:1008-1086)
    std::vector<se3d_person_msgs::Person2DList> people;
    while (ros::ok()) {
      {
        std::unique_lock<std::mutex> lck(g_mutex);
        g_cv.wait(lck, [] { return g_updated; });
        people = g_people;
        g_updated = false;
      }
      if (!ros::ok() || (int)people.size() != N) continue;
      se3d_person_msgs::PersonCovList persons3d;
      if (!node.skeletonThreadStep(people, persons3d)) continue;   // delta_t <= 0: frame skipped (S3D:1043-1046)
      se3d_visualization_msgs::MarkerArray markers;
      se3d::skeleton_markers(persons3d, prm.pose_method, markers);
      if (vis_cov) se3d::covariance_markers_3d(persons3d, prm.pose_method, markers, &sigma_range);
      pub_pers3d.publish(se3d_ros::to_ros(persons3d));
      pub_skel3d.publish(se3d_ros::to_ros(markers));
    }
  });

  se3d::Skeleton3DFrontEnd sync(N, [](const se3d::Skeleton3DFrontEnd::Tuple& tuple) {   // skeletonCallback (S3D:999-1006): latest wins
    {
      std::lock_guard<std::mutex> lck(g_mutex);
      g_people.clear();
      for (const auto& m : tuple) g_people.push_back(*m);
      g_updated = true;
    }
    g_cv.notify_one();
  });
  std::vector<ros::Subscriber> subs;
  for (int i = 0; i < N; ++i)
    subs.push_back(nh.subscribe<person_msgs::Person2DList>(cams[i] + "/human_joints", 1,
        [&sync, i](const person_msgs::Person2DList::ConstPtr& m) { sync.add(std::make_shared<const se3d_person_msgs::Person2DList>(se3d_ros::to_host(*m)), i); },
        ros::VoidConstPtr(), ros::TransportHints().tcpNoDelay()));
  ros::spin();

  {  // wake the worker so that it exits (S3D:1226-1231)
    std::lock_guard<std::mutex> lck(g_mutex);
    g_updated = true;
  }
  g_cv.notify_one();
  worker.join();
  se3d_ros::report_timings("Triangulation", node.timings_ms(), node.timing_counts());
  if (vis_cov) {
    std::cout << "Sigmas 3D [mm]:" << std::endl;   // S3D:1243-1247
    std::cout << "min: x: " << std::sqrt(sigma_range.min_ev[0]) * 1000 << "mm, y: " << std::sqrt(sigma_range.min_ev[1]) * 1000 << "mm, z: " << std::sqrt(sigma_range.min_ev[2]) * 1000 << "mm." << std::endl;
    std::cout << "max: x: " << std::sqrt(sigma_range.max_ev[0]) * 1000 << "mm, y: " << std::sqrt(sigma_range.max_ev[1]) * 1000 << "mm, z: " << std::sqrt(sigma_range.max_ev[2]) * 1000 << "mm." << std::endl;
  }
  return 0;
}
