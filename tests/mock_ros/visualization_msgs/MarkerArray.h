// Mock of <visualization_msgs/MarkerArray.h> (tests/mock_ros/README.md): visualization_msgs/Marker.msg's fields
#pragma once
#include <vector>
#include <geometry_msgs/Pose.h>
namespace visualization_msgs {
struct Marker {
  std_msgs::Header header;
  std::string ns;
  int32_t id = 0, type = 0, action = 0;
  geometry_msgs::Pose pose;
  geometry_msgs::Vector3 scale;
  std_msgs::ColorRGBA color;
  ros::Duration lifetime;
  uint8_t frame_locked = 0;
  std::vector<geometry_msgs::Point> points;
  std::vector<std_msgs::ColorRGBA> colors;
  std::string text, mesh_resource;
  uint8_t mesh_use_embedded_materials = 0;
};
struct MarkerArray { std::vector<Marker> markers; };
}
