#ifndef MOT_SHIM_OCCGRID_H
#define MOT_SHIM_OCCGRID_H
#include <std_msgs/Header.h>
#include <vector>
namespace geometry_msgs {
struct Point { double x = 0, y = 0, z = 0; };
struct Quaternion { double x = 0, y = 0, z = 0, w = 1; };
struct Pose { Point position; Quaternion orientation; };
struct Vector3 { double x = 0, y = 0, z = 0; };
}
namespace nav_msgs {
struct MapMetaData { ros::Time map_load_time; float resolution = 0; uint32_t width = 0, height = 0; geometry_msgs::Pose origin; };
struct OccupancyGrid { std_msgs::Header header; MapMetaData info; std::vector<int8_t> data; };
}
#endif
