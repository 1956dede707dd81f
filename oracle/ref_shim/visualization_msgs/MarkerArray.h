#ifndef MOT_SHIM_MARKERARRAY_H
#define MOT_SHIM_MARKERARRAY_H
#include <visualization_msgs/Marker.h>
namespace visualization_msgs { struct MarkerArray { std::vector<Marker> markers; }; }
namespace ros { namespace wire {
template <> struct Codec<visualization_msgs::MarkerArray> {
  static const char* type() { return "visualization_msgs/MarkerArray"; }
  static void write(Out& o, const visualization_msgs::MarkerArray& m) { o.msgs(m.markers); }
  static void read(In& i, visualization_msgs::MarkerArray& m) { i.msgs(m.markers); }
};
}}
#endif
