// Mock of <tf2_eigen/tf2_eigen.h>: tf2::transformToEigen (tests/mock_ros/README.md)
#pragma once
#include <Eigen/Core>
#include <geometry_msgs/Pose.h>
namespace tf2 {
inline Eigen::Isometry3d transformToEigen(const geometry_msgs::TransformStamped&) { return Eigen::Isometry3d(); }
}
