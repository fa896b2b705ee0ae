// Mock of <boost/shared_ptr.hpp> for the ROS mock: boost::shared_ptr as an alias of std::shared_ptr (tests/mock_ros/README.md).
#pragma once
#include <memory>
namespace boost {
template <class T> using shared_ptr = std::shared_ptr<T>;
}
