// person_msgs.hpp — plain C++ mirrors of the reference's ROS messages (person_msgs/msg/*.msg) for the ROS-free host layer.
// Field names and meaning follow the .msg files one to one, so a catkin build can swap these for the generated headers.
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <vector>

namespace person_msgs {

struct Header {            // std_msgs/Header (stamp as seconds)
  uint32_t seq = 0;
  double stamp = 0.0;
  std::string frame_id;
  // The ROS stamp this header was made from, in nanoseconds (-1: none).  The host layer computes with `stamp` (what the reference's
  // `.toSec()` yields) and only CARRIES this field along with copies of the header, so that a catkin wrapper publishes the input's sec / nsec
  // bit for bit instead of a double's ~240 ns grid (host/ros/ros_bridge.hpp; the reference forwards ros::Time unchanged, S3D:1060-1065).
  int64_t stamp_ns = -1;
};

struct Keypoint2D {        // Keypoint2D.msg
  float x = 0, y = 0, score = 0;
  std::array<float, 3> cov{{0, 0, 0}};  // xx, xy, yy
};

struct Person2D {          // Person2D.msg
  float score = 0;
  std::vector<Keypoint2D> keypoints;
  std::array<float, 4> bbox{{0, 0, 0, 0}};
};

struct Person2DList {      // Person2DList.msg
  Header header;
  float fb_delay = -1.f;
  std::vector<Person2D> persons;
};

struct Point { double x = 0, y = 0, z = 0; };  // geometry_msgs/Point

struct KeypointWithCovariance {  // KeypointWithCovariance.msg
  Point joint;
  float score = 0;
  std::array<double, 6> cov{{0, 0, 0, 0, 0, 0}};  // xx, xy, xz, yy, yz, zz
};

struct PersonCov {         // PersonCov.msg (bbox_center / bbox_size are never written by the nodes on this path)
  uint32_t id = 0;
  float score = 0;
  std::vector<KeypointWithCovariance> keypoints;
};

struct PersonCovList {     // PersonCovList.msg
  Header header;
  std::vector<double> ts_per_cam;
  std::vector<int64_t> ts_per_cam_ns;   // exact stamps alongside ts_per_cam (entries -1 / empty: none), carried like Header::stamp_ns
  std::vector<float> fb_delay_per_cam;
  std::vector<PersonCov> persons;
};

}  // namespace person_msgs
