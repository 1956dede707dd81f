// TEST INFRASTRUCTURE ONLY — ros::Time / ros::Duration of the file-backed mini-ROS (see ros/ros.h).
// Semantics follow roscpp's rostime: integer sec/nsec, fromSec() = floor + round of the fraction, toSec() = sec + 1e-9*nsec.
// Time::now() returns the replay clock that the input log sets ("__now__" records), so published stamps are reproducible.
#ifndef MOT_SHIM_ROS_TIME_H
#define MOT_SHIM_ROS_TIME_H
#include <cmath>
#include <cstdint>
namespace ros {
namespace shim { inline double& clock_now() { static double t = 0.0; return t; } }
struct Time {
  uint32_t sec = 0, nsec = 0;
  Time() {}
  Time(uint32_t s, uint32_t n) : sec(s), nsec(n) {}
  explicit Time(double t) { fromSec(t); }
  Time& fromSec(double t) {
    sec = (uint32_t)std::floor(t);
    nsec = (uint32_t)std::round((t - (double)sec) * 1e9);
    sec += nsec / 1000000000u; nsec %= 1000000000u;
    return *this;
  }
  Time& fromNSec(uint64_t t) { sec = (uint32_t)(t / 1000000000ull); nsec = (uint32_t)(t % 1000000000ull); return *this; }
  double toSec() const { return (double)sec + 1e-9 * (double)nsec; }
  uint64_t toNSec() const { return (uint64_t)sec * 1000000000ull + (uint64_t)nsec; }
  bool isZero() const { return sec == 0 && nsec == 0; }
  bool operator==(const Time& o) const { return sec == o.sec && nsec == o.nsec; }
  bool operator<(const Time& o) const { return sec < o.sec || (sec == o.sec && nsec < o.nsec); }
  static Time now() { return Time(shim::clock_now()); }
};
struct Duration {
  int32_t sec = 0, nsec = 0;
  Duration() {}
  Duration(int32_t s, int32_t n) : sec(s), nsec(n) {}
  Duration(double d) { fromSec(d); }   // implicit, like roscpp's explicit-less use `ros::Duration(0.1)`
  Duration& fromSec(double d) {
    int64_t s = (int64_t)std::floor(d);
    sec = (int32_t)s;
    nsec = (int32_t)std::round((d - (double)s) * 1e9);
    if (nsec >= 1000000000) { nsec -= 1000000000; sec += 1; }
    return *this;
  }
  double toSec() const { return (double)sec + 1e-9 * (double)nsec; }
};
}  // namespace ros
#endif
