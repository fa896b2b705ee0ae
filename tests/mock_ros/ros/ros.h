// Mock of <ros/ros.h> (roscpp): declarations only, for type-checking host/ros/*.cpp.  See tests/mock_ros/README.md.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include <boost/shared_ptr.hpp>

namespace ros {
struct Duration {
  int32_t sec = 0, nsec = 0;
  Duration() = default;
  explicit Duration(double t) { sec = (int32_t)std::floor(t); nsec = (int32_t)std::lround((t - sec) * 1e9); }
  double toSec() const { return (double)sec + 1e-9 * (double)nsec; }
  bool sleep() const { return true; }
};
struct Time {   // ros::Time: fromSec rounds to the nearest nanosecond (rostime: sec = floor(t), nsec = round((t - sec) 1e9))
  uint32_t sec = 0, nsec = 0;
  Time() = default;
  Time(uint32_t s, uint32_t ns) : sec(s), nsec(ns) {}
  explicit Time(double t) {
    sec = (uint32_t)std::floor(t);
    nsec = (uint32_t)std::lround((t - (double)sec) * 1e9);
    if (nsec >= 1000000000u) { sec += 1; nsec -= 1000000000u; }
  }
  double toSec() const { return (double)sec + 1e-9 * (double)nsec; }
  bool operator==(const Time& o) const { return sec == o.sec && nsec == o.nsec; }
  static Time now() { return Time(); }
};
struct Rate {
  explicit Rate(double) {}
  bool sleep() { return true; }
};
typedef boost::shared_ptr<void const> VoidConstPtr;
struct TransportHints {
  TransportHints& tcpNoDelay(bool = true) { return *this; }
};
struct Publisher {
  template <class M> void publish(const M&) const {}
};
struct Subscriber {
  Subscriber() {}
  ~Subscriber() {}   // (non-trivial like the real one: holding a subscriber is a use)
};
class NodeHandle {
 public:
  NodeHandle() = default;
  explicit NodeHandle(const std::string&) {}
  template <class T> bool param(const std::string&, T& val, const T& def) const { val = def; return false; }
  template <class T> T param(const std::string&, const T& def) const { return def; }
  template <class M> Publisher advertise(const std::string&, uint32_t, bool = false) { return Publisher(); }
  // the boost::function overload the wrappers use (lambdas): subscribe<M>(topic, queue, callback, tracked_object, transport_hints)
  template <class M>
  Subscriber subscribe(const std::string&, uint32_t, const std::function<void(const boost::shared_ptr<M const>&)>&, const VoidConstPtr& = VoidConstPtr(),
                       const TransportHints& = TransportHints()) {
    return Subscriber();
  }
};
inline void init(int&, char**, const std::string&) {}
inline bool ok() { return false; }
inline void spin() {}
inline void spinOnce() {}
}  // namespace ros

#define ROS_INFO(...) std::printf(__VA_ARGS__)
#define ROS_WARN(...) std::printf(__VA_ARGS__)
#define ROS_ERROR(...) std::printf(__VA_ARGS__)
#define ROS_FATAL(...) std::printf(__VA_ARGS__)
