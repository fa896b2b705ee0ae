// Mock of <sensor_msgs/CameraInfo.h> (tests/mock_ros/README.md): the fields of sensor_msgs/CameraInfo.msg the wrappers read
#pragma once
#include <array>
#include <vector>
#include <std_msgs/Header.h>
namespace sensor_msgs {
struct CameraInfo {
  std_msgs::Header header;
  uint32_t height = 0, width = 0;
  std::string distortion_model;
  std::vector<double> D;
  std::array<double, 9> K{};
  std::array<double, 9> R{};
  std::array<double, 12> P{};
  typedef boost::shared_ptr<CameraInfo const> ConstPtr;
};
}
