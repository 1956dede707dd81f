#ifndef MOT_SHIM_MARKER_H
#define MOT_SHIM_MARKER_H
#include <nav_msgs/OccupancyGrid.h>
#include <string>
#include <vector>
namespace std_msgs { struct ColorRGBA { float r = 0, g = 0, b = 0, a = 0; }; }
namespace visualization_msgs {
struct Marker {
  enum { ARROW = 0, CUBE = 1, SPHERE = 2, CYLINDER = 3, LINE_STRIP = 4, LINE_LIST = 5, CUBE_LIST = 6, SPHERE_LIST = 7, POINTS = 8 };
  enum { ADD = 0, MODIFY = 0, DELETE = 2 };
  std_msgs::Header header; std::string ns; int32_t id = 0; int32_t type = 0; int32_t action = 0;
  geometry_msgs::Pose pose; geometry_msgs::Vector3 scale; std_msgs::ColorRGBA color; ros::Duration lifetime;
  std::vector<geometry_msgs::Point> points;
};
}
#endif
