#ifndef MOT_SHIM_OBSTACLE_H
#define MOT_SHIM_OBSTACLE_H
// field list from /root/reference/object_tracking/msg/Obstacle.msg:1-8 (wire: 6 float64, int32, float64 — no padding)
#include <cstdint>
#include <ros/wire.h>
namespace object_tracking {
struct Obstacle { double x = 0, y = 0, z = 0, yaw = 0, pitch = 0, roll = 0; int32_t cluster = 0; double speed = 0; };
}
namespace ros { namespace wire {
template <> struct Codec<object_tracking::Obstacle> {
  typedef object_tracking::Obstacle M;
  static const char* type() { return "object_tracking/Obstacle"; }
  static void write(Out& o, const M& m) { o.pod(m.x); o.pod(m.y); o.pod(m.z); o.pod(m.yaw); o.pod(m.pitch); o.pod(m.roll); o.pod(m.cluster); o.pod(m.speed); }
  static void read(In& i, M& m) { i.pod(m.x); i.pod(m.y); i.pod(m.z); i.pod(m.yaw); i.pod(m.pitch); i.pod(m.roll); i.pod(m.cluster); i.pod(m.speed); }
};
}}
#endif
