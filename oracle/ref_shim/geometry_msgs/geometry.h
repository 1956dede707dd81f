// TEST INFRASTRUCTURE ONLY — the geometry_msgs value types used by the node sources (all float64 fields)
#ifndef MOT_SHIM_GEOMETRY_H
#define MOT_SHIM_GEOMETRY_H
#include <array>
#include <ros/wire.h>
namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 0; };
struct Pose { Point position; Quaternion orientation; };
struct Twist { Vector3 linear, angular; };
struct PoseWithCovariance { Pose pose; std::array<double, 36> covariance{}; };
struct TwistWithCovariance { Twist twist; std::array<double, 36> covariance{}; };
}
#endif
