// Mock of <tf2_ros/transform_listener.h> (+ tf2::TransformException) (tests/mock_ros/README.md)
#pragma once
#include <stdexcept>
#include <geometry_msgs/Pose.h>
namespace tf2 {
struct TransformException : std::runtime_error { using std::runtime_error::runtime_error; };
}
namespace tf2_ros {
struct Buffer {
  geometry_msgs::TransformStamped lookupTransform(const std::string&, const std::string&, const ros::Time&) const { return geometry_msgs::TransformStamped(); }
};
struct TransformListener { explicit TransformListener(Buffer&) {} };
}
