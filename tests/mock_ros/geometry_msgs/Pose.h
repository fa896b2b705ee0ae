// Mock of the geometry_msgs types the wrappers touch (tests/mock_ros/README.md)
#pragma once
#include <std_msgs/Header.h>
namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
struct Transform { Vector3 translation; Quaternion rotation; };
struct TransformStamped { std_msgs::Header header; std::string child_frame_id; Transform transform; };
}
