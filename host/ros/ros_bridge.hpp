// ros_bridge.hpp — glue between the generated ROS messages and the ROS-free host layer (host/se3d_nodes.hpp) for the three catkin node
// wrappers of this directory.  Compiled only inside a catkin workspace (host/ros/CMakeLists.txt: targets exist under find_package(catkin)).
//
// The host layer's message mirrors live in namespaces called person_msgs / visualization_msgs (host/person_msgs.hpp, host/se3d_markers.hpp)
// so that the ROS-free code reads like the reference.  In a ROS build those names belong to the generated headers, so the host layer is
// compiled into these targets with the two namespaces renamed (CMake: -Dperson_msgs=se3d_person_msgs -Dvisualization_msgs=se3d_visualization_msgs
// for the host sources) and this header includes the ROS headers FIRST, then the host headers under the same two macros.
#pragma once
#include <array>
#include <chrono>
#include <cmath>
#include <iostream>

#include <ros/ros.h>
#include <sensor_msgs/CameraInfo.h>
#include <tf2_eigen/tf2_eigen.h>
#include <tf2_ros/transform_listener.h>
#include <visualization_msgs/MarkerArray.h>

#include <person_msgs/Person2DList.h>
#include <person_msgs/PersonCovList.h>

#define person_msgs se3d_person_msgs
#define visualization_msgs se3d_visualization_msgs
#include "../se3d_markers.hpp"
#include "../se3d_nodes.hpp"
#undef person_msgs
#undef visualization_msgs

namespace se3d_ros {

const std::string BASE_FRAME = "base";                        // S3D:45, REPROJ:33
const std::string CAM_FRAME_SUFFIX = "_color_optical_frame";  // S3D:46
const std::string CAM_INFO_SUFFIX = "/color/camera_info";     // S3D:47

// private parameter "cameras" (S3D:1117-1130, REPROJ:246-258); the reference's built-in default is cam_1 .. cam_4 (S3D:49-52)
inline std::vector<std::string> camera_names(ros::NodeHandle& nh_private) {
  std::vector<std::string> names;
  nh_private.param("cameras", names, std::vector<std::string>());
  if (names.empty()) names = {"cam_1", "cam_2", "cam_3", "cam_4"};
  return names;
}

// getTransforms (S3D:160-192 / REPROJ:77-106): base -> camera optical frame, retried once per second until tf has all of them
inline std::vector<std::array<double, 16>> get_transforms(const std::vector<std::string>& cams, tf2_ros::Buffer& tfBuffer) {
  std::vector<std::array<double, 16>> out(cams.size());
  bool success = false;
  while (!success && ros::ok()) {
    try {
      for (size_t i = 0; i < cams.size(); ++i) {
        const auto tr = tfBuffer.lookupTransform(cams[i] + CAM_FRAME_SUFFIX, BASE_FRAME, ros::Time(0));
        const Eigen::Matrix<double, 4, 4, Eigen::RowMajor> M = tf2::transformToEigen(tr).matrix();
        std::copy(M.data(), M.data() + 16, out[i].begin());
      }
      success = true;
    } catch (tf2::TransformException& ex) {
      ROS_WARN("%s", ex.what());
      ros::Duration(1.0).sleep();
    }
  }
  ROS_INFO("Retrieved the camera extrinsic transforms.");
  return out;
}

// getIntrinsics (S3D:198-228 / REPROJ:112-137): one CameraInfo per camera, polled at 1 Hz until every camera has delivered
inline std::vector<sensor_msgs::CameraInfo> get_intrinsics(const std::vector<std::string>& cams, ros::NodeHandle& nh) {
  std::vector<sensor_msgs::CameraInfo> intr(cams.size());
  std::vector<char> got(cams.size(), 0);
  std::vector<ros::Subscriber> subs;
  for (size_t i = 0; i < cams.size(); ++i)
    subs.push_back(nh.subscribe<sensor_msgs::CameraInfo>(cams[i] + CAM_INFO_SUFFIX, 1, [&intr, &got, i](const sensor_msgs::CameraInfo::ConstPtr& m) { intr[i] = *m; got[i] = 1; }));
  ros::Rate rate(1.0);
  bool success = false;
  while (!success && ros::ok()) {
    ROS_INFO("Spinning.. Waiting to receive camera intrinsics.");
    ros::spinOnce();
    rate.sleep();
    success = true;
    for (size_t i = 0; i < cams.size(); ++i)
      if (!got[i] || (intr[i].D.size() == 0 && intr[i].distortion_model != "none")) { success = false; break; }   // S3D:216-221
  }
  ROS_INFO("intrinsics received.");
  return intr;
}

inline std::vector<se3d::CameraCalib> calibration(const std::vector<std::array<double, 16>>& T, const std::vector<sensor_msgs::CameraInfo>& intr) {
  std::vector<se3d::CameraCalib> cams(T.size());
  for (size_t i = 0; i < T.size(); ++i) {
    cams[i].T_base_to_cam = T[i];
    cams[i].fx = intr[i].K[0]; cams[i].fy = intr[i].K[4]; cams[i].cx = intr[i].K[2]; cams[i].cy = intr[i].K[5];   // S3D:314-317 (PinholeCameraModel fx .. cy)
    cams[i].width = (int)intr[i].width; cams[i].height = (int)intr[i].height;
  }
  return cams;
}

// ---- messages: generated ROS types <-> host mirrors (field for field, person_msgs/msg/*.msg) ----
// Stamps: the host layer computes with seconds as double (what the reference's `.toSec()` yields) and carries the message's sec / nsec next to
// it; a published stamp is the carried one whenever it still is the double's time (the reference forwards ros::Time unchanged, S3D:1060-1065,
// PP:528-532, REPROJ:230-231 — a double resolves ~240 ns at today's epoch, which would break stamp-equality matching downstream).
inline int64_t stamp_ns(const ros::Time& t) { return (int64_t)t.sec * 1000000000LL + (int64_t)t.nsec; }
inline ros::Time to_ros_time(double stamp, int64_t ns) {
  if (ns >= 0 && std::fabs((double)(ns / 1000000000LL) + 1e-9 * (double)(ns % 1000000000LL) - stamp) < 1e-6)
    return ros::Time((uint32_t)(ns / 1000000000LL), (uint32_t)(ns % 1000000000LL));
  return ros::Time(stamp);
}
inline se3d_person_msgs::Header to_host(const std_msgs::Header& h) {
  se3d_person_msgs::Header o;
  o.seq = h.seq; o.stamp = h.stamp.toSec(); o.stamp_ns = stamp_ns(h.stamp); o.frame_id = h.frame_id;
  return o;
}
inline std_msgs::Header to_ros(const se3d_person_msgs::Header& h) {
  std_msgs::Header o;
  o.seq = h.seq; o.stamp = to_ros_time(h.stamp, h.stamp_ns); o.frame_id = h.frame_id;
  return o;
}
inline se3d_person_msgs::Person2DList to_host(const person_msgs::Person2DList& m) {
  se3d_person_msgs::Person2DList o;
  o.header = to_host(m.header);
  o.fb_delay = m.fb_delay;
  o.persons.resize(m.persons.size());
  for (size_t p = 0; p < m.persons.size(); ++p) {
    const auto& s = m.persons[p];
    auto& d = o.persons[p];
    d.score = s.score;
    std::copy(s.bbox.begin(), s.bbox.end(), d.bbox.begin());
    d.keypoints.resize(s.keypoints.size());
    for (size_t k = 0; k < s.keypoints.size(); ++k) {
      d.keypoints[k].x = s.keypoints[k].x; d.keypoints[k].y = s.keypoints[k].y; d.keypoints[k].score = s.keypoints[k].score;
      std::copy(s.keypoints[k].cov.begin(), s.keypoints[k].cov.end(), d.keypoints[k].cov.begin());
    }
  }
  return o;
}
inline person_msgs::Person2DList to_ros(const se3d_person_msgs::Person2DList& m) {
  person_msgs::Person2DList o;
  o.header = to_ros(m.header);
  o.fb_delay = m.fb_delay;
  o.persons.resize(m.persons.size());
  for (size_t p = 0; p < m.persons.size(); ++p) {
    const auto& s = m.persons[p];
    auto& d = o.persons[p];
    d.score = s.score;
    std::copy(s.bbox.begin(), s.bbox.end(), d.bbox.begin());
    d.keypoints.resize(s.keypoints.size());
    for (size_t k = 0; k < s.keypoints.size(); ++k) {
      d.keypoints[k].x = s.keypoints[k].x; d.keypoints[k].y = s.keypoints[k].y; d.keypoints[k].score = s.keypoints[k].score;
      std::copy(s.keypoints[k].cov.begin(), s.keypoints[k].cov.end(), d.keypoints[k].cov.begin());
    }
  }
  return o;
}
inline se3d_person_msgs::PersonCovList to_host(const person_msgs::PersonCovList& m) {
  se3d_person_msgs::PersonCovList o;
  o.header = to_host(m.header);
  for (const auto& t : m.ts_per_cam) { o.ts_per_cam.push_back(t.toSec()); o.ts_per_cam_ns.push_back(stamp_ns(t)); }
  o.fb_delay_per_cam.assign(m.fb_delay_per_cam.begin(), m.fb_delay_per_cam.end());
  o.persons.resize(m.persons.size());
  for (size_t p = 0; p < m.persons.size(); ++p) {
    const auto& s = m.persons[p];
    auto& d = o.persons[p];
    d.id = s.id; d.score = s.score;
    d.keypoints.resize(s.keypoints.size());
    for (size_t k = 0; k < s.keypoints.size(); ++k) {
      d.keypoints[k].joint.x = s.keypoints[k].joint.x; d.keypoints[k].joint.y = s.keypoints[k].joint.y; d.keypoints[k].joint.z = s.keypoints[k].joint.z;
      d.keypoints[k].score = s.keypoints[k].score;
      std::copy(s.keypoints[k].cov.begin(), s.keypoints[k].cov.end(), d.keypoints[k].cov.begin());
    }
  }
  return o;
}
inline person_msgs::PersonCovList to_ros(const se3d_person_msgs::PersonCovList& m) {
  person_msgs::PersonCovList o;
  o.header = to_ros(m.header);
  for (size_t i = 0; i < m.ts_per_cam.size(); ++i)
    o.ts_per_cam.push_back(to_ros_time(m.ts_per_cam[i], i < m.ts_per_cam_ns.size() ? m.ts_per_cam_ns[i] : -1));
  o.fb_delay_per_cam.assign(m.fb_delay_per_cam.begin(), m.fb_delay_per_cam.end());
  o.persons.resize(m.persons.size());
  for (size_t p = 0; p < m.persons.size(); ++p) {
    const auto& s = m.persons[p];
    auto& d = o.persons[p];
    d.id = s.id; d.score = s.score;
    d.keypoints.resize(s.keypoints.size());
    for (size_t k = 0; k < s.keypoints.size(); ++k) {
      d.keypoints[k].joint.x = s.keypoints[k].joint.x; d.keypoints[k].joint.y = s.keypoints[k].joint.y; d.keypoints[k].joint.z = s.keypoints[k].joint.z;
      d.keypoints[k].score = s.keypoints[k].score;
      std::copy(s.keypoints[k].cov.begin(), s.keypoints[k].cov.end(), d.keypoints[k].cov.begin());
    }
  }
  return o;
}
inline visualization_msgs::MarkerArray to_ros(const se3d_visualization_msgs::MarkerArray& m) {
  visualization_msgs::MarkerArray o;
  o.markers.resize(m.markers.size());
  for (size_t i = 0; i < m.markers.size(); ++i) {
    const auto& s = m.markers[i];
    auto& d = o.markers[i];
    d.header = to_ros(s.header);
    d.ns = s.ns; d.id = s.id; d.type = s.type; d.action = s.action;
    d.pose.position.x = s.position.x; d.pose.position.y = s.position.y; d.pose.position.z = s.position.z;
    d.pose.orientation.x = s.orientation[0]; d.pose.orientation.y = s.orientation[1]; d.pose.orientation.z = s.orientation[2]; d.pose.orientation.w = s.orientation[3];
    d.scale.x = s.scale[0]; d.scale.y = s.scale[1]; d.scale.z = s.scale[2];
    d.color.r = s.color.r; d.color.g = s.color.g; d.color.b = s.color.b; d.color.a = s.color.a;
    d.lifetime = ros::Duration(s.lifetime);
    d.points.resize(s.points.size());
    for (size_t k = 0; k < s.points.size(); ++k) { d.points[k].x = s.points[k].x; d.points[k].y = s.points[k].y; d.points[k].z = s.points[k].z; }
    d.colors.resize(s.colors.size());
    for (size_t k = 0; k < s.colors.size(); ++k) { d.colors[k].r = s.colors[k].r; d.colors[k].g = s.colors[k].g; d.colors[k].b = s.colors[k].b; d.colors[k].a = s.colors[k].a; }
  }
  return o;
}

// "avg runtime" report at shutdown (S3D:1234-1241, PP:952-959)
inline void report_timings(const char* what, const std::vector<double>& ms, const std::vector<int>& cnt) {
  for (size_t i = 0; i < ms.size(); ++i)
    if (cnt[i] > 0) {
      std::cout << what << ": ";
      if (i > 0) std::cout << i << " detections: ";
      std::cout << "avg runtime: " << ms[i] / cnt[i] << "ms" << std::endl;
    }
}

}  // namespace se3d_ros
