// pose_prior_node.cpp — catkin wrapper of the accelerated pose_prior node: the ROS surface of the reference's main()
// (pose_prior/src/pose_prior_mult_node.cpp:923-960) around se3d::PosePriorNode (host/se3d_nodes.hpp -> libse3d_hip.so).
//   in : human_pose_estimation/persons_3d (person_msgs/PersonCovList, queue 1, tcpNoDelay)                                    PP:37, 947
//   out: human_pose_estimation/persons3d_fused, .../persons3d_fused_pred (PersonCovList), .../skeleton3d_fused (MarkerArray), queue 1   PP:943-945
//   private parameters: pose_method ("simple"), norm_height (false), vis_cov (false)                                           PP:930-932
// Single-threaded spinner with the track state inside the library's context, like the reference's global g_tracks (not re-entrant).
// The stage-2 kernels need no camera geometry, only the number of cameras (fb_delay_per_cam has one entry per camera): the context is
// created with placeholder calibrations for `cameras` (or n_cameras) views.
#include "ros_bridge.hpp"

int main(int argc, char** argv) {
  ros::init(argc, argv, "pose_prior");
  ros::NodeHandle nh;
  ros::NodeHandle nh_private("~");

  se3d::NodeParams prm;
  bool vis_cov = false;
  nh_private.param<std::string>("pose_method", prm.pose_method, "simple");
  nh_private.param<bool>("norm_height", prm.norm_height, false);
  nh_private.param<bool>("vis_cov", vis_cov, false);
  int device = 0, max_tracks = 64, n_cameras = (int)se3d_ros::camera_names(nh_private).size();
  nh_private.param<int>("device", device, 0);              // (new) HIP device of this node
  nh_private.param<int>("max_tracks", max_tracks, 64);     // (new) initial limit; the context grows when detections + tracks exceed it
  nh_private.param<int>("n_cameras", n_cameras, n_cameras);
  ROS_INFO("Using pose method %s (norm_height = %d) and skeleton model with limb-length sigma factor of %f.", prm.pose_method.c_str(), prm.norm_height,
           prm.norm_height ? 2.0 : 1.0);   // PP:934-939

  std::vector<se3d::CameraCalib> placeholder((size_t)std::max(2, n_cameras));
  for (size_t i = 0; i < placeholder.size(); ++i) {   // distinct poses so that the (unused) rig tensors are well defined
    placeholder[i].T_base_to_cam = {1, 0, 0, (double)i, 0, 1, 0, 0, 0, 0, 1, 5, 0, 0, 0, 1};
    placeholder[i].fx = placeholder[i].fy = 600.0; placeholder[i].cx = 320.0; placeholder[i].cy = 240.0;
  }
  std::shared_ptr<se3d::Context> ctx;
  try {
    ctx = std::make_shared<se3d::Context>(placeholder, prm, 16, max_tracks, device);
  } catch (const std::exception& e) {
    ROS_FATAL("%s", e.what());
    return -1;
  }
  se3d::PosePriorNode node(ctx);
  se3d::FusedMarkerState marker_state;
  std::vector<double> timings(10, 0.0);
  std::vector<int> timing_cnt(10, 0);

  ros::Publisher pub_fusion = nh.advertise<person_msgs::PersonCovList>("human_pose_estimation/persons3d_fused", 1);
  ros::Publisher pub_fusion_pred = nh.advertise<person_msgs::PersonCovList>("human_pose_estimation/persons3d_fused_pred", 1);
  ros::Publisher pub_fusion_marker = nh.advertise<visualization_msgs::MarkerArray>("human_pose_estimation/skeleton3d_fused", 1);

  ros::Subscriber sub_person3d = nh.subscribe<person_msgs::PersonCovList>("human_pose_estimation/persons_3d", 1,
      [&](const person_msgs::PersonCovList::ConstPtr& persons) {   // skeletonCallback (PP:505-921)
        const auto t1 = std::chrono::high_resolution_clock::now();
        se3d_person_msgs::PersonCovList fused, fused_pred;
        node.skeletonCallback(se3d_ros::to_host(*persons), fused, fused_pred);
        se3d_visualization_msgs::MarkerArray markers;
        marker_state.update(fused, markers);        // DELETE markers of vanished tracks (PP:191-209, 878-891)
        se3d::fused_markers(fused, markers);
        if (vis_cov) se3d::covariance_markers_fused(fused, markers);
        pub_fusion_marker.publish(se3d_ros::to_ros(markers));   // publish order of PP:905-907
        pub_fusion.publish(se3d_ros::to_ros(fused));
        pub_fusion_pred.publish(se3d_ros::to_ros(fused_pred));
        const double ms = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::high_resolution_clock::now() - t1).count() / 1000.0;
        timings[0] += ms; ++timing_cnt[0];   // PP:912-920
        if (fused.persons.size() > 0 && fused.persons.size() < timings.size()) { timings[fused.persons.size()] += ms; ++timing_cnt[fused.persons.size()]; }
      },
      ros::VoidConstPtr(), ros::TransportHints().tcpNoDelay());
  ros::spin();
  se3d_ros::report_timings("Skeleton Model", timings, timing_cnt);
  return 0;
}
