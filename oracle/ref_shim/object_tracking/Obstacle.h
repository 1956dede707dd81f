#ifndef MOT_SHIM_OBSTACLE_H
#define MOT_SHIM_OBSTACLE_H
// field list from /root/reference/object_tracking/msg/Obstacle.msg:1-8
#include <cstdint>
namespace object_tracking {
struct Obstacle { double x = 0, y = 0, z = 0, yaw = 0, pitch = 0, roll = 0; int32_t cluster = 0; double speed = 0; };
}
#endif
