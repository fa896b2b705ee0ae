// Mock of <std_msgs/Header.h> / <std_msgs/ColorRGBA.h> (tests/mock_ros/README.md)
#pragma once
#include <string>
#include <ros/ros.h>
namespace std_msgs {
struct Header { uint32_t seq = 0; ros::Time stamp; std::string frame_id; };
struct ColorRGBA { float r = 0, g = 0, b = 0, a = 0; };
}
