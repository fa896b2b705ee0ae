// pose_reprojection_node.cpp — catkin wrapper of the accelerated feedback-reprojection node: the ROS surface of the reference's main()
// (pose_reprojection/src/skeleton_reproj_mult_node.cpp:237-296) around se3d::ReprojectionNode (host/se3d_nodes.hpp -> libse3d_hip.so).
//   in : human_pose_estimation/persons3d_fused_pred (person_msgs/PersonCovList, queue 1, tcpNoDelay)    REPROJ:41, 293
//   out: <cam>/skel_pred (person_msgs/Person2DList, queue 1), one topic per camera                      REPROJ:35, 281-284
//   private parameters: pose_method ("simple"), cameras ([]: cam_1 .. cam_4)                            REPROJ:244-258
#include "ros_bridge.hpp"

int main(int argc, char** argv) {
  ros::init(argc, argv, "multi_skeleton_reprojection");
  ros::NodeHandle nh;
  ros::NodeHandle nh_private("~");

  se3d::NodeParams prm;
  nh_private.param<std::string>("pose_method", prm.pose_method, "simple");
  int device = 0, max_persons = 64;
  nh_private.param<int>("device", device, 0);               // (new) HIP device of this node
  nh_private.param<int>("max_persons", max_persons, 64);    // (new) capacity in persons per message
  const std::vector<std::string> cams = se3d_ros::camera_names(nh_private);
  const int N = (int)cams.size();
  ROS_INFO("NUM_CAMERAS: %d", N);
  ROS_INFO("Using pose estimation method: %s, and %d keypoints.", prm.pose_method.c_str(), 17);

  tf2_ros::Buffer tfBuffer;
  tf2_ros::TransformListener tfListener(tfBuffer);
  const auto T = se3d_ros::get_transforms(cams, tfBuffer);
  const auto intr = se3d_ros::get_intrinsics(cams, nh);
  std::vector<ros::Publisher> skeleton_pubs;
  for (int i = 0; i < N; ++i) skeleton_pubs.push_back(nh.advertise<person_msgs::Person2DList>(cams[i] + "/skel_pred", 1));
  ROS_INFO("Reprojecting into %zu / %d camera views", T.size(), N);
  if (T.size() != intr.size() || T.size() != skeleton_pubs.size() || N < 2) {
    ROS_ERROR("incoherent number of transforms, intrinsics and output heatmaps! Aborting!");   // REPROJ:287-290
    return -1;
  }
  std::shared_ptr<se3d::Context> ctx;
  try {
    ctx = std::make_shared<se3d::Context>(se3d_ros::calibration(T, intr), prm, 16, max_persons, device);
  } catch (const std::exception& e) {
    ROS_FATAL("%s", e.what());
    return -1;
  }
  se3d::ReprojectionNode node(ctx);   // uploads the double-precision poses, projection parameters and image sizes (REPROJ:77-137)

  ros::Subscriber fused_skel_sub = nh.subscribe<person_msgs::PersonCovList>("human_pose_estimation/persons3d_fused_pred", 1,
      [&](const person_msgs::PersonCovList::ConstPtr& humans) {   // fusedSkeletonCallback (REPROJ:139-235)
        std::vector<se3d_person_msgs::Person2DList> humans_in_cam;
        if (!node.fusedSkeletonCallback(se3d_ros::to_host(*humans), humans_in_cam)) return;   // wrong frame_id / ts_per_cam size (REPROJ:141-147)
        for (int i = 0; i < N && i < (int)humans_in_cam.size(); ++i) skeleton_pubs[i].publish(se3d_ros::to_ros(humans_in_cam[i]));
      },
      ros::VoidConstPtr(), ros::TransportHints().tcpNoDelay());
  ros::spin();
  return 0;
}
