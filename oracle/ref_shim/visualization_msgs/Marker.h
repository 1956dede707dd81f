// TEST INFRASTRUCTURE ONLY — visualization_msgs/Marker with the full ROS1 field list (so the wire bytes are the real ones)
#ifndef MOT_SHIM_MARKER_H
#define MOT_SHIM_MARKER_H
#include <std_msgs/Header.h>
#include <geometry_msgs/geometry.h>
#include <string>
#include <vector>
namespace visualization_msgs {
struct Marker {
  enum { ARROW = 0, CUBE = 1, SPHERE = 2, CYLINDER = 3, LINE_STRIP = 4, LINE_LIST = 5, CUBE_LIST = 6, SPHERE_LIST = 7, POINTS = 8,
         TEXT_VIEW_FACING = 9, MESH_RESOURCE = 10, TRIANGLE_LIST = 11 };
  enum { ADD = 0, MODIFY = 0, DELETE = 2, DELETEALL = 3 };
  std_msgs::Header header; std::string ns; int32_t id = 0; int32_t type = 0; int32_t action = 0;
  geometry_msgs::Pose pose; geometry_msgs::Vector3 scale; std_msgs::ColorRGBA color; ros::Duration lifetime;
  uint8_t frame_locked = 0;
  std::vector<geometry_msgs::Point> points; std::vector<std_msgs::ColorRGBA> colors;
  std::string text, mesh_resource; uint8_t mesh_use_embedded_materials = 0;
};
}
namespace ros { namespace wire {
template <> struct Codec<visualization_msgs::Marker> {
  typedef visualization_msgs::Marker M;
  static const char* type() { return "visualization_msgs/Marker"; }
  static void write(Out& o, const M& m) {
    o.msg(m.header); o.str(m.ns); o.pod(m.id); o.pod(m.type); o.pod(m.action); o.pod(m.pose); o.pod(m.scale); o.pod(m.color);
    o.duration(m.lifetime); o.pod(m.frame_locked); o.pods(m.points); o.pods(m.colors); o.str(m.text); o.str(m.mesh_resource);
    o.pod(m.mesh_use_embedded_materials);
  }
  static void read(In& i, M& m) {
    i.msg(m.header); i.str(m.ns); i.pod(m.id); i.pod(m.type); i.pod(m.action); i.pod(m.pose); i.pod(m.scale); i.pod(m.color);
    i.duration(m.lifetime); i.pod(m.frame_locked); i.pods(m.points); i.pods(m.colors); i.str(m.text); i.str(m.mesh_resource);
    i.pod(m.mesh_use_embedded_materials);
  }
};
}}
#endif
