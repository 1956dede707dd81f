// TEST INFRASTRUCTURE ONLY — nav_msgs/OccupancyGrid (Header, MapMetaData{time, float32 resolution, uint32 width, height, Pose}, int8[] data)
#ifndef MOT_SHIM_OCCGRID_H
#define MOT_SHIM_OCCGRID_H
#include <std_msgs/Header.h>
#include <geometry_msgs/geometry.h>
#include <vector>
namespace nav_msgs {
struct MapMetaData { ros::Time map_load_time; float resolution = 0; uint32_t width = 0, height = 0; geometry_msgs::Pose origin; };
struct OccupancyGrid { std_msgs::Header header; MapMetaData info; std::vector<int8_t> data; };
}
namespace ros { namespace wire {
template <> struct Codec<nav_msgs::OccupancyGrid> {
  static const char* type() { return "nav_msgs/OccupancyGrid"; }
  static void write(Out& o, const nav_msgs::OccupancyGrid& m) {
    o.msg(m.header); o.time(m.info.map_load_time); o.pod(m.info.resolution); o.pod(m.info.width); o.pod(m.info.height);
    o.pod(m.info.origin); o.pods(m.data);
  }
  static void read(In& i, nav_msgs::OccupancyGrid& m) {
    i.msg(m.header); i.time(m.info.map_load_time); i.pod(m.info.resolution); i.pod(m.info.width); i.pod(m.info.height);
    i.pod(m.info.origin); i.pods(m.data);
  }
};
}}
#endif
